#!/bin/bash
# r06 session 22: the rule of equal-length slices for the two-word LDS-offset builds (rb_phase_plan.h, phase_multi_equal_slices) on the GPU:
# parity (the whole -m gpu suite; test_several_reads_per_wave now cuts into 3 / 7 / 31 equal slices too), the planner guard over the
# two-word points the rule changed and over its default 16, the merged shapes of the bench around the rule's window, the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06u
mkdir -p $OUT
cd $R
( time timeout 1500 python3 -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -6 $OUT/pytest_gpu.txt
timeout 1500 python3 profiles/phase_rule_check.py --points 2:200:16,2:300:16,2:200:18.9,2:250:18.9,2:300:18.9,2:360:18.9,2:250:22,2:360:22,2:200:26,2:250:26,2:300:26,2:360:26,2:250:31,2:360:31 > $OUT/guard_two_word_equal.txt 2>&1
echo "guard (two-word, equal slices) exit $?"; grep -v "^      " $OUT/guard_two_word_equal.txt | cut -c1-260
timeout 1500 python3 profiles/phase_rule_check.py > $OUT/guard_default.txt 2>&1
echo "guard (default points) exit $?"; grep -v "^      " $OUT/guard_default.txt | cut -c1-260
export RB_TUNING_ENV=1
timeout 900 python3 profiles/multi_reads_sweep.py --workloads deplete_target,targets3,deplete_target360,targets3_360 --rpw 1 --skew 2 --factors 0.8,0.86,0.92,0.96,1.0,1.04,1.08,1.15,1.25 2>&1 | grep -v amdgpu.ids | tee $OUT/merged_shapes_rule.txt | cut -c1-330
unset RB_TUNING_ENV
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default_line.json 2> $OUT/bench_default.err
wc -c $OUT/bench_default_line.json; tail -3 $OUT/bench_default.err
cp bench_detail.json $OUT/bench_default.json
python3 - <<'PY'
import json
d = json.load(open("/root/repo/gpurun_out/r06u/bench_default.json"))
print(d["value"], d["roofline"]["frac"])
for k, l in d["other_configs"].items():
    r = l.get("roofline") or {}
    print("  ", k, l.get("value"), r.get("avg_kernel_ms"), r.get("frac"), (r.get("request_bound") or {}).get("request_bound_frac"))
PY
echo done
