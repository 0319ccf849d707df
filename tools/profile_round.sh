#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>   -> gpurun_out/<tag>/...
# The rocprofv3 runs behind profiles/<tag>_*: kernel-trace stats of the cfg3 and cfg4 bench commands, the HBM traffic
# counters (separate --pmc passes), the SQ counters of the bound sheet, and the plain default line.
set -e
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 20 --warmup 5 > $out/bench_default_line.json 2> $out/bench_default.err
echo "default line done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_cfg3 -- python3 bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_cfg3_line.json 2> $out/kt_cfg3.err
echo "cfg3 trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_cfg4 -- python3 bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_cfg4_line.json 2> $out/kt_cfg4.err
echo "cfg4 trace done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_$c.log 2>&1
done
echo "traffic done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $out/pmc_sq1 -- python3 bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq2 -- python3 bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_sq2.log 2>&1
python3 tools/pmc_summary.py $out/pmc_sq1 $out/pmc_sq2 > $out/pmc_sq_summary.txt
python3 tools/make_traffic_json.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/traffic_cfg3.json > /dev/null
find $out/kt_cfg3 -name "*kernel_stats.csv" -exec cp {} $out/bench_cfg3_kernel_stats.csv \;
find $out/kt_cfg4 -name "*kernel_stats.csv" -exec cp {} $out/bench_cfg4_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_cfg5r -- python3 bench.py --workload cfg5r --steps 2 --warmup 1 --corpus 5000 --no-cpu-baseline > $out/bench_cfg5r_line.json 2> $out/kt_cfg5r.err
find $out/kt_cfg5r -name "*kernel_stats.csv" -exec cp {} $out/bench_cfg5r_kernel_stats.csv \;
echo "cfg5r trace done"
python3 tools/len_sweep.py 4000000 8000 16000 24000 32000 40000 48000 65000 65536 80000 131072 > $out/len_sweep.txt 2>&1
F2CNN_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 2 --steps 5 --warmup 2 --corpus 2000 > $out/bench_2rank_one_device_line.json 2> $out/bench_2rank.err
# CNN kernels: kernel-trace statistics + three SQ counter passes of the same command -> bound sheet
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_k4 -- python3 tools/k4_probe.py 14240 > $out/k4_probe.log 2>&1
find $out/kt_k4 -name "*kernel_stats.csv" -exec cp {} $out/k4_kernel_stats.csv \;
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $out/pmc_k4_a -- python3 tools/k4_probe.py 14240 > $out/pmc_k4_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_k4_b -- python3 tools/k4_probe.py 14240 > $out/pmc_k4_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VMEM --output-format csv -d $out/pmc_k4_c -- python3 tools/k4_probe.py 14240 > $out/pmc_k4_c.log 2>&1
python3 tools/pmc_summary.py $out/pmc_k4_a $out/pmc_k4_b $out/pmc_k4_c > $out/pmc_sq_k4.txt
python3 tools/cnn_bound_sheet.py $out/pmc_sq_k4.txt $out/k4_kernel_stats.csv > $out/cnn_bound_sheet.txt 2>&1 || true
echo "cnn counters done"
# HBM traffic of the ragged pass (per launch and per row of the long-row kernel)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc5r_$c -- python3 bench.py --workload cfg5r --corpus 2500 --steps 1 --warmup 1 --no-cpu-baseline > $out/pmc5r_$c.log 2>&1
done
python3 tools/make_traffic_cfg5r.py $out/pmc5r_FETCH_SIZE $out/pmc5r_WRITE_SIZE $out/traffic_cfg5r.json 2500 > /dev/null 2>&1 || true
echo "cfg5r traffic done"
./tools/ubench/fma64_operands > $out/ubench_fma64.txt 2>&1 || true
./tools/ubench/mfma_valu_coissue > $out/ubench_mfma_valu_coissue.txt 2>&1 || true
./tools/ubench/mfma_fillers > $out/ubench_mfma_fillers.txt 2>&1 || true
timeout -k 10 300 python3 -m pytest tests/test_gpu_spectral.py -q -s -k speech_shaped > $out/guard_speech_shaped.txt 2>&1 || true
echo "all done"
