"""GPU: the drop-in executables (wenet_amd/bin/{fsk_demod,drs232_ldpc,wenet_ldpc}) honour the reference's
pipe contract: `fsk_demod --cu8 -s --stats=100 M Fs Rs - - 2> stats | {drs232,wenet}_ldpc - - -v`
(start_rx.sh:125-128, benchmarking/test_demod.py:26-43) gives the bytes the reference pipeline gives."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from conftest import load_golden
from wenet_amd import siggen

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "wenet_amd", "bin")
FMT_FLAG = {"cu8": "--cu8", "cs16": "--cs16", "s16": ""}


def run_pipe(g, tmp_path, soft=True, stats=True):
    cfg = siggen.CONFIGS[str(g["config"])]()
    raw = tmp_path / "cap.bin"
    g["raw"].tofile(str(raw))
    l2 = "drs232_ldpc" if cfg.mode == 1 else "wenet_ldpc"
    st = tmp_path / "stats.txt"
    l2err = tmp_path / "l2.txt"
    cmd = (f"cat {raw} | {BIN}/fsk_demod {FMT_FLAG[str(g['fmt'])]} -s {'--stats=100' if stats else ''} {cfg.M} {cfg.Fs} {cfg.Rs} - - 2> {st} "
           f"| {BIN}/{l2} - - -v 2> {l2err}")
    out = subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, check=True).stdout
    return out, open(st).read(), open(l2err).read(), cfg


@pytest.mark.parametrize("name", ["v1_8dB", "v2_8dB", "v2_cs16_10dB", "v2_ppm150_12dB", "4fsk_12dB", "v2_6dB"])
def test_shell_pipeline_matches_reference(name, tmp_path):
    g = load_golden(name)
    out, stats, l2err, cfg = run_pipe(g, tmp_path)
    assert out == g["packets"].tobytes()
    iters = [int(l.split("iter:")[1]) for l in l2err.splitlines() if "iter:" in l]
    assert iters == list(g["iters"])
    last = l2err.strip().splitlines()[-1]
    n_all, n_bad = g["iters"].size, g["iters"].size - g["packets"].size // 256
    assert last.startswith(f"packets: {n_all} packet_errors: {n_bad} PER:")
    # stderr JSON schema + values (src/fsk_demod.c:351-392); consumers: rx/fskstatsudp.py:29, rx/fskdemodgui.py
    lines = [l for l in stats.splitlines() if l.startswith("{")]
    assert lines
    for mine, ref, frame in ((lines[0], str(g["stats_first"]), 1),):
        a, b = json.loads(mine), json.loads(ref)
        assert set(a) == set(b)
        for k in a:
            if k == "secs":
                continue
            if k == "eye_diagram":
                assert np.array(a[k]).shape == np.array(b[k]).shape
                high = math.ceil(float(g["trace"][frame, 5]) * (cfg.Fs // cfg.Rs))
                if high >= -1:          # otherwise the reference indexes f_int[] out of bounds (fsk.c:1045,1060)
                    assert a[k] == b[k]
                continue
            assert a[k] == b[k], k
    # number of JSON lines = frames printed on the same schedule as the reference
    # schedule: frames 1, 1+(stats_loop+1), ...  (src/fsk_demod.c:247-251, 345-401)
    loop_time = np.float32(cfg.Ts * 48) / np.float32(cfg.Fs)
    stats_loop = int(1 / (100 * loop_time))
    nframes = g["trace"].shape[0]
    assert len(lines) == len(range(1, nframes, stats_loop + 1))
    b_last = json.loads(str(g["stats_last"]))
    a_last = json.loads(lines[-1])
    for k in ("EbNodB", "ppm", "f1_est", "f2_est", "samp_fft"):
        assert a_last[k] == b_last[k], k


def test_hard_decision_output_and_files(tmp_path):
    g = load_golden("v1_20dB")
    cfg = siggen.CONFIGS["v1"]()
    raw = tmp_path / "cap.bin"; g["raw"].tofile(str(raw))
    bits = tmp_path / "bits.bin"
    subprocess.run([f"{BIN}/fsk_demod", "--cu8", str(cfg.M), str(cfg.Fs), str(cfg.Rs), str(raw), str(bits)], check=True,
                   stderr=subprocess.DEVNULL)
    assert (np.packbits(np.fromfile(str(bits), np.uint8)) == g["hard"]).all()
    sd = tmp_path / "sd.bin"
    subprocess.run([f"{BIN}/fsk_demod", "-d", "-s", "-p", str(cfg.Ts), str(cfg.M), str(cfg.Fs), str(cfg.Rs), str(raw), str(sd)], check=True,
                   stderr=subprocess.DEVNULL)
    assert (np.fromfile(str(sd), np.float32).view(np.uint32) == g["sd"].view(np.uint32)).all()
    pk = tmp_path / "pk.bin"
    subprocess.run([f"{BIN}/drs232_ldpc", str(sd), str(pk)], check=True, stderr=subprocess.DEVNULL)
    assert open(pk, "rb").read() == g["packets"].tobytes()


def test_cli_usage_and_errors_equal_the_reference_binaries(tmp_path):
    """Same stderr text and exit status as the reference executables for the argument errors a script can hit (the program
    name in the usage line is argv[0], so both are run through a link of the same name)."""
    if not ol.have_ref():
        pytest.skip("oracle/_ref not built")
    dirs = {}
    for tag, d in (("ref", ol.REF_DIR), ("gpu", BIN)):
        dd = tmp_path / tag
        dd.mkdir()
        for exe in ("fsk_demod", "drs232_ldpc", "wenet_ldpc"):
            os.symlink(os.path.join(d, exe), dd / exe)
        dirs[tag] = dd
    cases = [["fsk_demod", "2", "960000"], ["fsk_demod", "-h"], ["fsk_demod", "2", "960000", "96000", "-", "-", "extra"],
             ["fsk_demod", "3", "960000", "96000", "-", "-"], ["fsk_demod", "2", "960000", "96000", "/nonexistent/x", "-"],
             ["wenet_ldpc", "-"], ["drs232_ldpc"], ["drs232_ldpc", "/nonexistent/x", "-"], ["wenet_ldpc", "-", "/nonexistent/dir/y"]]
    for c in cases:
        res = []
        for tag in ("ref", "gpu"):
            r = subprocess.run(["./" + c[0]] + c[1:], cwd=dirs[tag], stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            res.append((r.returncode, r.stdout, r.stderr))
        assert res[0] == res[1], c


def test_cli_error_behaviour(tmp_path):
    r = subprocess.run([f"{BIN}/fsk_demod", "2", "960000"], stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Too few arguments" in r.stderr and b"usage:" in r.stderr
    r = subprocess.run([f"{BIN}/fsk_demod", "3", "960000", "96000", "-", "-"], stderr=subprocess.PIPE, stdin=subprocess.DEVNULL)
    assert r.returncode == 1 and b"Mode 3 is not valid" in r.stderr
    r = subprocess.run([f"{BIN}/fsk_demod", "2", "960000", "96000", "/nonexistent/x", "-"], stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Couldn't open files" in r.stderr
    r = subprocess.run([f"{BIN}/wenet_ldpc", "-"], stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"usage: drs232" in r.stderr
    r = subprocess.run([f"{BIN}/drs232_ldpc", "/nonexistent/x", "-"], stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Error opening input file" in r.stderr
    # empty input: exit 0, nothing written, summary line printed (PER of 0/0 as the reference prints it)
    r = subprocess.run([f"{BIN}/wenet_ldpc", "-", "-"], stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == b"" and b"packets: 0 packet_errors: 0" in r.stderr


@pytest.mark.parametrize("flags", [["-f"], ["-f", "-s"], ["-f", "--stats=50"]])
def test_testframe_mode_matches_reference(flags, tmp_path):
    """fsk_demod -f (src/fsk_demod.c:226-245,304-343): sliding compare against the known 100-bit frame (srand(158324)),
    'errs: ...' lines or, with -t, one JSON line per frame with a detection.  Same stderr as the reference binary."""
    import ctypes as C
    import json
    import re
    if not ol.have_ref():
        pytest.skip("oracle/_ref not built")
    libc = C.CDLL("libc.so.6")
    libc.srand(158324)
    frame = np.array([libc.rand() & 1 for _ in range(100)], np.uint8)
    cfg = siggen.config_v2()
    rng = np.random.default_rng(9)
    bits = np.concatenate([rng.integers(0, 2, 777, dtype=np.uint8), np.tile(frame, 40), rng.integers(0, 2, 300, dtype=np.uint8), np.tile(frame, 9)])
    x = siggen.add_noise(siggen.modulate(bits, cfg), cfg, 9.0, rng)
    raw = tmp_path / "tf.cu8"
    siggen.to_cu8(x).tofile(raw)
    outs = []
    for exe in (os.path.join(ol.REF_DIR, "fsk_demod"), os.path.join(BIN, "fsk_demod")):
        p = subprocess.run([exe, "--cu8"] + flags + [str(cfg.M), str(cfg.Fs), str(cfg.Rs), str(raw), str(tmp_path / "o.bin")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        outs.append((p.stderr.decode(), (tmp_path / "o.bin").read_bytes()))
    (ref_err, ref_out), (my_err, my_out) = outs
    assert my_out == ref_out
    strip = lambda t: re.sub(r'"secs": \d+', '"secs": 0', t)
    assert strip(my_err) == strip(ref_err)
    assert ("errs:" in ref_err) or ('"frames"' in ref_err)                  # the pattern was really found
    if "--stats=50" in flags:
        last = json.loads([l for l in my_err.splitlines() if l.startswith("{")][-1])
        assert last["frames"] >= 40 and last["bits"] == 100 * last["frames"]


def test_sigterm_exits_zero():
    """src/fsk_demod.c:47-52,264: SIGTERM while waiting for input ends the process with exit status 0; what was
    demodulated before has been written (stdout is flushed per block)."""
    import signal
    import time
    cfg = siggen.config_v2()
    raw, _ = siggen.make_capture(cfg, 2, 12.0, seed=3)
    p = subprocess.Popen([f"{BIN}/fsk_demod", "--cu8", "-s", str(cfg.M), str(cfg.Fs), str(cfg.Rs), "-", "-"],
                         stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    p.stdin.write(raw.tobytes()); p.stdin.flush()
    want = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M)[0].tobytes()
    got = b""
    t0 = time.time()
    os.set_blocking(p.stdout.fileno(), False)
    while len(got) < len(want) and time.time() - t0 < 60:
        chunk = p.stdout.read()
        if chunk:
            got += chunk
        else:
            time.sleep(0.05)
    p.send_signal(signal.SIGTERM)                         # stdin still open: the process is blocked in read()
    assert p.wait(timeout=30) == 0
    assert got == want


@pytest.mark.parametrize("name", ["v2_8dB", "v1_8dB", "v2_ppm150_12dB"])
def test_fused_executable_gives_the_pipe_bytes(name, tmp_path):
    """wenet_rx = fsk_demod | {drs232,wenet}_ldpc in ONE process: same packets on stdout, the L2 tools' stderr lines."""
    g = load_golden(name)
    cfg = siggen.CONFIGS[str(g["config"])]()
    raw = tmp_path / "cap.bin"
    g["raw"].tofile(str(raw))
    r = subprocess.run(f"cat {raw} | {BIN}/wenet_rx {FMT_FLAG[str(g['fmt'])]} -m {cfg.mode} -v {cfg.M} {cfg.Fs} {cfg.Rs} - -", shell=True,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    assert r.stdout == g["packets"].tobytes()
    iters = [int(l.split("iter:")[1]) for l in r.stderr.decode().splitlines() if "iter:" in l]
    assert iters == list(g["iters"])
    out = tmp_path / "pk.bin"
    subprocess.run([f"{BIN}/wenet_rx", FMT_FLAG[str(g["fmt"])], f"--framing={cfg.mode}", str(cfg.M), str(cfg.Fs), str(cfg.Rs), str(raw), str(out)],
                   check=True, stderr=subprocess.DEVNULL)
    assert out.read_bytes() == g["packets"].tobytes()


def test_get_demod_stats_mirror():
    """fsk_get_demod_stats (src/fsk.h:130) through the C ABI: after each fsk_demod_sd call the stats are those of that frame
    (the values the CLI prints are checked against the reference's JSON in test_shell_pipeline_matches_reference)."""
    from wenet_amd.fsk import Fsk
    cfg = siggen.config_v2()
    x, _ = siggen.make_capture(cfg, 1, 12.0, seed=6, fmt="cf32")
    f = Fsk(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M)
    st0 = f.get_demod_stats()
    assert st0.nfft_est == 0 and st0.snr_est == 0.0
    f.enable_stats(0, 1)
    off = 0
    seen = []
    for _ in range(12):
        n = f.nin()
        f.demod_sd(x[off:off + n]); off += n
        st = f.get_demod_stats()
        seen.append((st.snr_est, tuple(st.f_est)[:2], st.rx_timing))
        assert st.nfft_est == 128 and st.neyesamp > 0
    snap = f.get_stats(4)
    assert len(snap) == 1 and snap[0].snr_est == seen[-1][0]
    assert abs(seen[-1][1][1] - seen[-1][1][0] - cfg.Rs) < 2 * cfg.Fs / 256            # tones one symbol rate apart, to a bin
    assert len({s[0] for s in seen}) > 6                                                # the figures move from frame to frame
    f.close()
