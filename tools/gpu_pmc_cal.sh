# Calibration of the VALU utilisation figure: tools/ubench/pmc_cal (known instruction streams saturating every SIMD) bare and under rocprofv3 --pmc.
# Writes gpurun_out/r03_valu_calibration.json (copy to profiles/).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
./tools/ubench/pmc_cal > $OUT/pmc_cal_bare.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  n=$(echo $set | md5sum | cut -c1-6)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_cal_$n -o p -- $GRAFT_REPO_ROOT/tools/ubench/pmc_cal > $OUT/pmc_cal_$n.log 2>&1
done
python - <<'PY'
import csv, glob, json, os, re
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
bare = open(root + "/pmc_cal_bare.txt").read()
acc = {}
for f in glob.glob(root + "/pmc_cal_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        grid = int(r.get("Grid_Size", 0) or 0)
        key = k + ("/1w" if grid and grid <= 64 else "")
        d = acc.setdefault(key, {})
        d[r["Counter_Name"]] = max(d.get(r["Counter_Name"], 0.0), float(r["Counter_Value"]))
out = {"note": "tools/ubench/pmc_cal.hip under rocprofv3 --pmc (tools/gpu_pmc_cal.sh): known instruction streams, 2048 x 512 threads = 8 wavefronts per SIMD, "
               "20000 trips of 64 operations per wave (1w: one wave alone, 160000 trips).  per_inst = counter / wave-instructions issued.",
       "bare_run": bare.splitlines(), "kernels": {}}
for k, d in acc.items():
    one = k.endswith("/1w")
    insts = (1 if one else 2048 * 8) * (160000 if one else 20000) * 64
    e = dict(d); e["wave_instructions"] = insts
    e["per_inst"] = {c: v / insts for c, v in d.items()}
    if d.get("GRBM_GUI_ACTIVE"):
        simd_cycles = d["GRBM_GUI_ACTIVE"] / 8 * 1024
        e["simd_cycles_per_inst"] = simd_cycles / insts
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_INST_CYCLES_VALU", "SQ_BUSY_CYCLES"):
            if c in d: e[c + "_x4_over_simd_cycles"] = d[c] * 4 / simd_cycles
    out["kernels"][k] = e
json.dump(out, open(root + "/r03_valu_calibration.json", "w"), indent=1)
print(bare)
for k, e in out["kernels"].items():
    print(k, {c: round(v, 4) for c, v in e["per_inst"].items()}, "simd cyc/inst", round(e.get("simd_cycles_per_inst", 0), 3))
PY
