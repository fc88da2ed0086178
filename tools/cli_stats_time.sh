# development aid: wall time of the fsk_demod drop-in alone, with and without the stats side channel, against the reference binary
cd $GRAFT_REPO_ROOT
python - <<'PY'
import subprocess, sys, time
sys.path.insert(0, '.')
from wenet_amd import siggen
cfg = siggen.config_v2()
raw, _ = siggen.make_capture(cfg, 340, 8.0, seed=3)
raw.tofile('/tmp/c.cu8')
print(raw.size / 2 / cfg.Fs, 's of signal')
for exe in ("wenet_amd/bin/fsk_demod", "oracle/_ref/fsk_demod"):
    for s in ([], ["--stats=100"], ["--stats=8"]):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            subprocess.run([exe, "--cu8", "-s"] + s + ["2", "960000", "96000", "/tmp/c.cu8", "/dev/null"], stderr=subprocess.DEVNULL)
            ts.append(time.perf_counter() - t0)
        print(exe, s, " ".join(f"{t:.3f}" for t in ts))
PY
