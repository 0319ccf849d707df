#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in main p3; do
  if [ $v = p3 ]; then export F2CNN_PROBE_LIB=tools/libf2cnn_hip_p3.so; else unset F2CNN_PROBE_LIB; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_plan_$v -- python3 tools/k2_probe.py 1000 > gpurun_out/pmc_plan_$v.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for v in ("main", "p3"):
    f = glob.glob(f"gpurun_out/pmc_plan_{v}/**/*counter_collection.csv", recursive=True)[0]
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_envelope<float, 13>" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    print(v)
    for d in sorted(rows):
        c = rows[d]
        print("  dispatch %4d  WAVE %.3e  WAIT_ANY %.3e  WAIT_INST %.3e  ACTIVE %.3e  VALU %.3e LDS %.3e WAIT_LDS %.3e  INSTS_VALU %.3e" % (d, c.get("SQ_WAVE_CYCLES",0), c.get("SQ_WAIT_ANY",0), c.get("SQ_WAIT_INST_ANY",0), c.get("SQ_ACTIVE_INST_ANY",0), c.get("SQ_ACTIVE_INST_VALU",0), c.get("SQ_ACTIVE_INST_LDS",0), c.get("SQ_WAIT_INST_LDS",0), c.get("SQ_INSTS_VALU",0)))
PY
