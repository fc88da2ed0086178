"""GPU: link compatibility.  oracle/_ref/{fsk_demod,drs232_ldpc,wenet_ldpc}_on_shim are the reference's OWN mains -- the unmodified
src/fsk_demod.c, src/drs232_ldpc.c, src/wenet_ldpc.c compiled against the reference's headers -- linked with wenet_amd/libwenet_fsk_compat.so instead
of the reference's fsk.c / kiss_fft.c / mpdecode_core.c / phi0.c (oracle/Makefile, built where /root/reference exists; the binaries travel).  Their
stdout and stderr must equal the pure reference binaries' byte for byte: struct FSK / MODEM_STATS / LDPC layouts, names and semantics all hold."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from conftest import load_golden
from wenet_amd import siggen

pytestmark = pytest.mark.gpu
REF = ol.REF_DIR
FMT_FLAG = {"cu8": ["--cu8"], "cs16": ["--cs16"], "s16": []}


def _have():
    return all(os.path.exists(os.path.join(REF, b + s)) for b in ("fsk_demod", "drs232_ldpc", "wenet_ldpc") for s in ("", "_on_shim"))


@pytest.mark.skipif(not _have(), reason="oracle/_ref/*_on_shim not built (make -C oracle ref, where /root/reference exists)")
@pytest.mark.parametrize("name", ["v2_8dB", "v1_8dB", "4fsk_12dB", "v2_cs16_10dB"])
def test_reference_mains_linked_with_the_compat_library_equal_the_reference_binaries(name, tmp_path):
    g = load_golden(name)
    cfg = siggen.CONFIGS[str(g["config"])]()
    raw = ol.raw_bytes(g["raw"]).tobytes()
    args = FMT_FLAG[str(g["fmt"])] + ["-s", "--stats=7", str(cfg.M), str(cfg.Fs), str(cfg.Rs), "-", "-"]
    outs = {}
    for tag in ("", "_on_shim"):
        p = subprocess.run([os.path.join(REF, "fsk_demod" + tag)] + args, input=raw, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, timeout=600)
        outs[tag] = p
    assert outs["_on_shim"].stdout == outs[""].stdout                              # every soft decision
    assert len(outs[""].stdout) == g["sd"].size * 4

    def stats_lines(b):                                                           # stderr: one JSON line per 8th frame; "secs" is wall-clock time
        import json
        rows = []
        for l in b.decode().splitlines():
            if l.startswith("{"):
                d = json.loads(l); d.pop("secs", None); rows.append(d)
        return rows
    a, b = stats_lines(outs["_on_shim"].stderr), stats_lines(outs[""].stderr)
    assert len(a) == len(b) > 0
    for x, y in zip(a, b):
        x.pop("eye_diagram", None); y.pop("eye_diagram", None)                     # (the reference reads f_int[] out of bounds for some timings, fsk.c:1045,1060: tests/test_gpu_cli.py)
        assert x == y
    # hard decisions through fsk_demod()
    hb = [subprocess.run([os.path.join(REF, "fsk_demod" + tag)] + FMT_FLAG[str(g["fmt"])] + [str(cfg.M), str(cfg.Fs), str(cfg.Rs), "-", "-"], input=raw,
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True, timeout=600).stdout for tag in ("", "_on_shim")]
    assert hb[0] == hb[1] and len(hb[0]) == g["sd"].size
    # the second stage: sd_to_llr + run_ldpc_decoder through the reference's own deframer loop
    l2 = "drs232_ldpc" if cfg.mode == 1 else "wenet_ldpc"
    res = [subprocess.run([os.path.join(REF, l2 + tag), "-", "-", "-v"], input=outs[""].stdout, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, timeout=600)
           for tag in ("", "_on_shim")]
    assert res[1].stdout == res[0].stdout == g["packets"].tobytes()
    assert res[1].stderr == res[0].stderr                                          # "packets: .. packet_errors: .. PER: .. iter: .." per packet + the summary
