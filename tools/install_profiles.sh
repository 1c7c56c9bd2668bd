#!/bin/bash
# usage: tools/install_profiles.sh <rNN>  -- copies gpurun_out/final_<rNN>/ (tools/final_round.sh) into profiles/<rNN>_* and derives the traffic JSONs
r=$1
F=gpurun_out/final_$r
H=$(git log --format=%h -1 -- carla-driving-rl-agent_amd/csrc)
cp $F/bench.json profiles/${r}_bench.json
cp $F/kernel_trace_summary.md profiles/${r}_kernel_trace_summary.md
cp $F/timeline_step.txt profiles/${r}_timeline_step.txt
rm -rf profiles/${r}_pmc profiles/${r}_c3
mkdir -p profiles/${r}_pmc profiles/${r}_c3
cp $F/pmc/*.txt profiles/${r}_pmc/
cp $F/pmc_mfma.json profiles/${r}_pmc_mfma.json
cp $F/final_configs.txt profiles/${r}_final_configs.txt
cp $F/rollout_rows.json profiles/${r}_rollout_rows.json
cp $F/c3/bench_lines.txt $F/c3/kernel_trace_summary.md $F/c3/step_FETCH_SIZE.txt $F/c3/step_WRITE_SIZE.txt $F/c3/timeline_step.txt profiles/${r}_c3/
cp $F/pmc_sq.json profiles/${r}_pmc_sq.json
cp $F/timeline_idle.txt profiles/${r}_timeline_idle.txt
rm -rf profiles/${r}_w360 && mkdir -p profiles/${r}_w360 && cp $F/w360/* profiles/${r}_w360/
python tools/pmc_traffic_json.py $F/w360 "kernels at commit $H; 90x360 (F5); calibration passes of profiles/${r}_pmc" 256 4 90 360 f32 $F/pmc > profiles/${r}_w360_pmc_traffic.json
python tools/pmc_traffic_json.py $F/pmc "kernels at commit $H" > profiles/${r}_pmc_traffic.json
python tools/pmc_traffic_json.py $F/c3 "kernels at commit $H; calibration passes of profiles/${r}_pmc" 1024 4 90 120 bf16s $F/pmc > profiles/${r}_c3_bf16s_pmc_traffic.json
