"""Identity of the kernel sources a measurement belongs to.

A PMC profile under profiles/ describes ONE build of the kernels.  tools/gpu_profile_round.sh stamps `source_sha16()` into the profile
it writes, and bench.py quotes a profile's traffic / VALU figures only if the stamp equals the sources it runs from -- a profile that
was taken before the last kernel change drops out of the bench line (`roofline.traffic` null) instead of describing other code.
The hash covers the device sources and the build flags, not the binary: a rebuild of the same sources keeps it."""
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wenet_amd", "csrc")
# everything that is compiled into the demodulator / decoder kernels (the host-only files wenet_rx.hip, cli_*.cpp are not)
_KERNEL_SOURCES = ("Makefile", "demod_common.h", "wenet_internal.h", "glibc_atan2f.h", "x87emu.h", "demod_oct_impl.h", "demod_oct.hip", "demod_oct_sliced.hip",
                   "demod_pipe_impl.h", "demod_tri_impl.h", "demod_chain_split.h", "demod_pipe_arrive.inc", "demod_pipe_shared_1.inc", "demod_pipe_shared_2.inc",
                   "demod_pipe_shared_3.inc", "demod_pipe_shared_4.inc", "demod_pipe_kernel.hip", "demod_pipe_raw.hip", "demod_pipe_tri.hip",
                   "demod_kernel.hip", "ldpc_kernel.hip", "ldpc_host_tables.h", "tables/ldpc_vpos.inc")


def source_sha16():
    h = hashlib.sha256()
    for name in _KERNEL_SOURCES:
        p = os.path.join(CSRC, name)
        h.update(name.encode() + b"\0")
        if os.path.exists(p):
            h.update(open(p, "rb").read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def library_source_id():
    """source_sha16() of the sources libwenet_rx.so was BUILT from (the Makefile compiles it in: wenet_rx_source_id()).  A profile or a bench
    line may speak for the current sources only if this equals source_sha16() -- otherwise the library is a stale build."""
    from . import lib
    L = lib.load()
    return L.wenet_rx_source_id().decode()


def library_sha16():
    p = os.path.join(ROOT, "wenet_amd", "libwenet_rx.so")
    return hashlib.sha256(open(p, "rb").read()).hexdigest()[:16] if os.path.exists(p) else None


def hipcc_version():
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        for line in out.splitlines():
            if "HIP version" in line:
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


if __name__ == "__main__":
    print(source_sha16())
