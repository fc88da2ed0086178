# development aid: kernel names + LDS/VGPR/scratch of our dispatches (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kinfo -o k -- python $GRAFT_REPO_ROOT/bench.py --captures ${1:-768} --steps 1 --warmup 0 --no-cpu-baseline --no-single-stream > /tmp/kinfo.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/kinfo/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "wenet" in r["Kernel_Name"]:
            print(r["Kernel_Name"][:60], "grid", r["Grid_Size_X"], "wg", r["Workgroup_Size_X"], "lds", r["LDS_Block_Size"], "scratch", r["Scratch_Size"],
                  "vgpr", r["VGPR_Count"], "agpr", r["Accum_VGPR_Count"], "sgpr", r["SGPR_Count"], "ms", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
PY
