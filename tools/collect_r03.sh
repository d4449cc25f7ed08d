#!/bin/bash
# Round-3 evidence on one MI355X box: bench line, kernel statistics, PMC passes (separate runs, as the guide prescribes), the
# rank-3-of-8 shard, configs 4 and 5.  Everything goes to gpurun_out/r03/ and is copied into profiles/r03/ afterwards.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o sweep -- python3 $R/tools/profile_sweep.py 1024 512 10 > $O/prof.log 2>&1 || exit 1
echo stats done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 $R/tools/profile_sweep.py 1024 512 3 > $O/pmc_w.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 $R/tools/profile_sweep.py 1024 512 3 > $O/pmc_f.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- python3 $R/tools/profile_sweep.py 1024 512 3 > $O/pmc_sq.log 2>&1 || exit 1
echo pmc done
rocprofv3 --kernel-trace --stats -d $O/prof_shard -o shard -- python3 $R/tools/profile_shard.py 8 3 10 > $O/prof_shard.log 2>&1 || exit 1
cd $R
python tools/rocpd_stats.py $O/prof/sweep_results.db $O/bench_kernel_stats.csv > $O/kernel_stats.txt
python tools/rocpd_stats.py $O/prof_shard/shard_results.db $O/shard_8way_rank3_kernel_stats.csv > /dev/null
python tools/pmc_traffic.py $O/pmc_w/w_counter_collection.csv $O/pmc_f/f_counter_collection.csv $O/pmc_traffic.json > /dev/null
cp $O/pmc_traffic.json $R/profiles/r03/pmc_traffic.json      # bench.py reads it there (and refuses a file of other sources)
python bench.py > $O/bench_r03.json 2> $O/bench_r03.err || exit 1
TMF_BENCH_SAME_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-sample 0 > $O/bench_r03_2ranks_one_gpu.json 2> $O/bench_2.err || exit 1
echo bench done
python tools/pmc_sq.py $O/pmc_sq/sq_counter_collection.csv $O/pmc_mfma_lds.json > /dev/null 2>&1
python tools/run_cfg4.py --reps 6 > $O/cfg4_kitaev_L512_chi256.log 2>&1
python tools/run_cfg4.py --random --reps 6 > $O/cfg4_random_bdg_L512_chi256.log 2>&1
python tools/run_cfg5.py --reps 3 --json $O/cfg5_gutzwiller_parallel.json > $O/cfg5_gutzwiller_parallel.log 2>&1
python tools/run_cfg5.py --reps 3 --method sequential > $O/cfg5_gutzwiller_sequential.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c5 -- python3 $R/tools/run_cfg5.py --reps 3 > $O/prof_c5.log 2>&1 || exit 1
cd $R
python tools/rocpd_stats.py $O/prof_c5/c5_results.db $O/cfg5_gutzwiller_kernel_stats.csv > /dev/null
python tools/shard_host_cost.py > $O/shard_host_cost_8ranks.log 2>&1
python tools/shard_host_cost.py 0:292 292:512 0:512 >> $O/shard_host_cost_8ranks.log 2>&1
echo collected
