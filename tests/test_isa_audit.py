"""CPU: the built code objects obey the rule round 5 paid for -- every s_barrier is preceded, on every path, by a completed wait for the wavefront's LDS stores
(tools/isa_barrier_audit.py).  hipcc 7.2 dropped that wait at the barrier heading the decoder's packet loop (the stores arrive over the loop's back edges), and on gfx950 a
read behind the barrier then overtook the store about once in 10^6 packets; the kernels now write the wait out (WR_LDS_BARRIER / lds_barrier), and this test keeps it so."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_barrier_is_reached_by_an_lds_store_in_flight():
    objs = sorted(glob.glob(os.path.join(ROOT, "wenet_amd", "csrc", "*.o")))
    assert objs, "build first (python -c 'import __graft_entry__ as g; g.build()')"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_barrier_audit.py")] + objs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.endswith(" 0 findings") and int(last.split()[0]) > 200, last        # (297 barriers in the round-5 build)


def test_the_audit_sees_the_dropped_wait_in_the_minimal_kernel(tmp_path):
    """tools/ubench/syncthreads_loop_header.hip is the decoder's packet loop reduced to 45 lines: hipcc 7.2 emits its loop-header barrier without the wait for the claim store.
    The audit must find exactly that (if a later compiler keeps the wait, the finding is gone and this test says so by skipping)."""
    import pytest
    obj = str(tmp_path / "slh.o")
    c = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-c", os.path.join(ROOT, "tools", "ubench", "syncthreads_loop_header.hip"), "-o", obj],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert c.returncode == 0, c.stdout[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_barrier_audit.py"), obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    if r.returncode == 0:
        pytest.skip("this compiler keeps the wait at the loop header: " + r.stdout.strip().splitlines()[-1])
    assert "ds_write_b32" in r.stdout and "offset:39184" in r.stdout, r.stdout[-2000:]


def test_m0_of_the_decoder_is_written_by_its_add_tid_blocks_alone():
    """The check pass of wenet_decode_kernel stores through ds_write_addtid_b32 (address = M0 + offset + 4 lane) and writes M0 once per check, not in front of every store
    (ldpc_kernel.hip).  That holds only while nothing else in the kernel touches M0 between the write and the stores; the compiler does not know the asm blocks read M0, so
    the code object is checked: every mention of m0 is `s_mov_b32 m0, s<n>` (ours), every add-TID store lies behind one, and an `s_nop` follows each write."""
    import re
    import shutil
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    obj = os.path.join(ROOT, "wenet_amd", "csrc", "ldpc_kernel.o")
    assert os.path.exists(obj), "build first"
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, "ldpc_kernel.o")
        shutil.copy(obj, local)
        subprocess.run([f"{llvm}/llvm-objdump", "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        cos = sorted(glob.glob(local + ".*gfx950*"))
        assert cos, "no gfx950 code object in ldpc_kernel.o"
        dis = subprocess.run([f"{llvm}/llvm-objdump", "-d", cos[0]], stdout=subprocess.PIPE, text=True, check=True).stdout
    m = re.search(r"<_Z19wenet_decode_kernel12WrDecodeArgs>:\n(.*?)\n\n", dis, re.S)
    assert m, "wenet_decode_kernel not found"
    lines = [ln.split("//")[0].strip() for ln in m.group(1).splitlines() if ln.strip()]
    addtid = [i for i, ln in enumerate(lines) if ln.startswith("ds_write_addtid_b32")]
    m0 = [i for i, ln in enumerate(lines) if re.search(r"\bm0\b", ln)]
    if not addtid:
        assert not m0
        return
    assert m0 and m0[0] < addtid[0], "an add-TID store in front of the first write of M0"
    for i in m0:
        assert re.fullmatch(r"s_mov_b32 m0, s\d+", lines[i]), f"M0 touched by: {lines[i]}"
        assert lines[i + 1].startswith("s_nop"), f"no wait state behind the write of M0: {lines[i + 1]}"
    assert len({lines[i] for i in m0}) == 1, "M0 written from different registers"
