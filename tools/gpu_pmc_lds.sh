# development aid: LDS activity / bank-conflict counters of our kernels at the bench workload -> gpurun_out/<tag>_lds_counters.txt
# usage: gpu_pmc_lds.sh [captures] [tag]
cd /tmp && export TMPDIR=/tmp
B=${1:-3584}; TAG=${2:-r04}; shift; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_lds -o s -- python $GRAFT_REPO_ROOT/bench.py --captures $B --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > /tmp/pmc_lds.log 2>&1
python - $B <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/${TAG}_lds_counters.txt
import csv, glob, sys
print("# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -- python bench.py --captures %s --steps 1 (tools/gpu_pmc_lds.sh)" % sys.argv[1])
print("# per kernel: the launch with the largest SQ_LDS_IDX_ACTIVE.  SQ_LDS_IDX_ACTIVE = LDS-array cycles summed over the 256 CUs, SQ_LDS_BANK_CONFLICT = the conflict cycles among them;")
print("# GRBM_GUI_ACTIVE is summed over the 8 XCDs: launch cycles = GRBM_GUI_ACTIVE / 8; lds_array_busy = SQ_LDS_IDX_ACTIVE / (256 x launch cycles)")
for f in glob.glob("/tmp/pmc_lds/*counter_collection.csv"):
    per = {}
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "wenet_demod" not in k and "wenet_decode" not in k and "wenet_llr" not in k: continue
        per.setdefault((k.split("(")[0][:60], r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    best = {}
    for (k, d), c in per.items():
        if k not in best or c.get("SQ_LDS_IDX_ACTIVE", 0) > best[k].get("SQ_LDS_IDX_ACTIVE", 0): best[k] = c
    for k, c in sorted(best.items()):
        cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
        idx, bc, n = c.get("SQ_LDS_IDX_ACTIVE", 0), c.get("SQ_LDS_BANK_CONFLICT", 0), c.get("SQ_INSTS_LDS", 0)
        print(k)
        print("   " + "  ".join("%s %.4g" % kv for kv in sorted(c.items())))
        if cyc and n:
            print("   launch %.3g cycles; lds_array_busy %.3f; conflict share of the array cycles %.3f; per LDS wave-instruction: %.2f array cycles of which %.2f conflicts" % (cyc, idx / (256 * cyc), bc / idx if idx else 0, idx / n, bc / n))
PY
tail -2 /tmp/pmc_lds.log | cut -c1-200
