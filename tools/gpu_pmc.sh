# Development aid: one rocprofv3 --pmc pass (never with a trace domain) over bench.py at a batch, per-kernel counter values printed and kept.
#   usage: gpu_pmc.sh "<counters>|<preset>" [captures] [tag] [extra bench args]      PMC_ENV="WENET_RX_OCT=7" forces library switches for the run
#   presets (the sets rounds 2-4 used, one script each then): wait = where the wavefronts' cycles go | insts = instruction mix | icache | sq = VALU / LDS activity
# (gpu_pmc_lds.sh and gpu_pmc_cal.sh keep their own post-processing; tools/gpu_profile_round.sh is the committed per-round profile.)
cd /tmp && export TMPDIR=/tmp
SET=$1; B=${2:-3584}; TAG=${3:-r05}; shift; shift; shift
case "$SET" in
  wait)   SET="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM";;
  insts)  SET="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC";;
  icache) SET="SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY";;
  sq)     SET="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_BUSY_CYCLES";;
esac
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
n=$(echo $SET | md5sum | cut -c1-6)
env $PMC_ENV rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_${TAG}_$n -o p -- python $GRAFT_REPO_ROOT/bench.py --captures $B --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > $OUT/pmc_${TAG}_$n.log 2>&1
python - $OUT/pmc_${TAG}_$n "$SET" <<'PY' | tee $OUT/${TAG}_pmc_$n.txt
import csv, glob, sys
print("# rocprofv3 --pmc", sys.argv[2], "(tools/gpu_pmc.sh); per kernel: the largest value over its dispatches")
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "wenet" not in k: continue
        d = acc.setdefault(k.split("(")[0][:70], {})
        d[r["Counter_Name"]] = max(d.get(r["Counter_Name"], 0.0), float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    wc = d.get("SQ_WAVE_CYCLES")
    print(k)
    for c, v in sorted(d.items()): print(f"    {c:28s} {v:.5g}" + (f"   {v / wc:.3f} of wave cycles" if wc else ""))
PY
