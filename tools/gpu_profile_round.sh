# One profiling round on the GPU box (outputs under gpurun_out/, to be copied into profiles/):
#   rocprofv3 --kernel-trace --stats of the bench command | PMC passes (FETCH_SIZE, WRITE_SIZE, three SQ sets; never with trace domains) | the bench line, last
# usage: gpu_profile_round.sh [captures] [tag] [extra bench args...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B=${1:-3584}; TAG=${2:-r05}; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o k -- python $GRAFT_REPO_ROOT/bench.py --captures $B --steps 5 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/${TAG}_prof.log 2>&1
cp $OUT/prof_$TAG/k_kernel_stats.csv $OUT/${TAG}_kernel_stats_b$B.csv
for pass in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
            "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VALU"; do
  d=$OUT/pmc_${TAG}_$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py --captures $B --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > $d.log 2>&1
done
python - "$B" "$TAG" "$@" <<'PY'
import csv, glob, json, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from wenet_amd import codeid
B = int(sys.argv[1]); tag = sys.argv[2]
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
secs = 10.0
if "--seconds" in sys.argv: secs = float(sys.argv[sys.argv.index("--seconds") + 1])
cfg = "v2"
if "--config" in sys.argv: cfg = sys.argv[sys.argv.index("--config") + 1]
Fs = {"v2": 960000, "v1": 921416, "4fsk": 1843200}[cfg]
N = {"v2": 480, "v1": 384, "4fsk": 1536}[cfg]
NS = B * int(secs * Fs)
if codeid.library_source_id() != codeid.source_sha16():
    sys.exit(f"libwenet_rx.so was built from {codeid.library_source_id()}, the tree holds {codeid.source_sha16()}: a profile of a stale build is not written")
acc = {}
for f in glob.glob(f"{root}/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "wenet" not in k: continue
        d = acc.setdefault(k[:64], {})
        d[r["Counter_Name"]] = max(d.get(r["Counter_Name"], 0.0), float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc, one pass per counter set (FETCH_SIZE | WRITE_SIZE | SQ set), tools/gpu_profile_round.sh; bench.py --steps 1 --warmup 0 at the "
               "bench batch.  FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section; calibrated there for wide "
               "streaming reads -- this kernel reads 2..4 bytes per lane, so the corrected figure is an upper bound); WRITE_SIZE uncorrected.  valu_busy = "
               "SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); packed-f32 instructions occupy a SIMD twice as long.",
       "valu_note": "valu_util = (plain VALU x 2.16 + packed-f32 VALU x 4.12 cycles) / SIMD cycles of the launch (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), the per-instruction "
                    "cycles of a saturated SIMD from profiles/r03_valu_calibration.json (tools/ubench/pmc_cal.hip); packed = SQ_INSTS_VALU_FLOPS_FP32 - ADD_F32 - MUL_F32 - "
                    "2 FMA_F32 (a packed add / multiply counts two flops and one instruction there).  It cannot exceed 1.  The round-2 figure valu_busy (every instruction "
                    "taken as 4 cycles) is kept beside it for comparison.",
       "source_sha16": codeid.source_sha16(), "library_sha16": codeid.library_sha16(), "library_source_id": codeid.library_source_id(),
       "captures": B, "samples_in_launch": NS, "kernels": {}}
for k, d in acc.items():
    e = dict(d)
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        e["read_bytes_corrected"] = 2 * 1024 * d.get("FETCH_SIZE", 0.0)
        e["write_bytes"] = 1024 * d.get("WRITE_SIZE", 0.0)
        e["hbm_bytes"] = e["read_bytes_corrected"] + e["write_bytes"]
        e["hbm_bytes_per_iq_sample"] = e["hbm_bytes"] / NS
    if d.get("GRBM_GUI_ACTIVE"):
        simd_cycles = d["GRBM_GUI_ACTIVE"] / 8 * 1024
        e["valu_busy"] = round(d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / simd_cycles, 4)
        e["lds_busy"] = round(d.get("SQ_ACTIVE_INST_LDS", 0) * 4 / simd_cycles, 4)
        if d.get("SQ_ACTIVE_INST_VALU"): e["lanes_active"] = round(d.get("SQ_THREAD_CYCLES_VALU", 0) / d["SQ_ACTIVE_INST_VALU"], 2)
        if d.get("SQ_ACTIVE_INST_LDS"): e["lds_bank_conflict_ratio"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_ACTIVE_INST_LDS"], 3)
        if "demod" in k: e["valu_insts_per_frame"] = round(d.get("SQ_INSTS_VALU", 0) / (NS / N), 1)
        if "SQ_INSTS_VALU_FLOPS_FP32" in d and d.get("SQ_INSTS_VALU"):
            packed = max(0.0, d["SQ_INSTS_VALU_FLOPS_FP32"] - d.get("SQ_INSTS_VALU_ADD_F32", 0) - d.get("SQ_INSTS_VALU_MUL_F32", 0) - 2 * d.get("SQ_INSTS_VALU_FMA_F32", 0))
            packed = min(packed, d["SQ_INSTS_VALU"])
            e["valu_packed_share"] = round(packed / d["SQ_INSTS_VALU"], 4)
            e["valu_util"] = round(((d["SQ_INSTS_VALU"] - packed) * 2.16 + packed * 4.12) / simd_cycles, 4)
            e["simd_cycles_per_valu_inst"] = round(((d["SQ_INSTS_VALU"] - packed) * 2.16 + packed * 4.12) / d["SQ_INSTS_VALU"], 3)
        if d.get("SQ_WAVE_CYCLES"):
            e["wave_cycles_share"] = {c[3:].lower(): round(d[c] / d["SQ_WAVE_CYCLES"], 4) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS") if c in d}
    out["kernels"][k] = e
json.dump(out, open(f"{root}/{tag}_pmc_b{B}.json", "w"), indent=1)
print(json.dumps({k[:40]: {x: v[x] for x in ("hbm_bytes_per_iq_sample", "valu_busy", "valu_util", "valu_packed_share", "lanes_active", "valu_insts_per_frame", "lds_bank_conflict_ratio", "wave_cycles_share") if x in v} for k, v in out["kernels"].items() if "demod" in k or "decode" in k}, indent=1))
PY
# the bench line LAST: it quotes traffic / valu from the PMC profile written just now (same sources: the stamp is checked)
cd $GRAFT_REPO_ROOT
python bench.py --captures $B "$@" 2> $OUT/${TAG}_bench.err | tail -1 > $OUT/${TAG}_bench_b$B.json
head -12 $OUT/${TAG}_kernel_stats_b$B.csv | cut -c1-160
tail -c 2500 $OUT/${TAG}_bench_b$B.json
