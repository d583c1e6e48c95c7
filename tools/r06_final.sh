#!/bin/bash
# round 6, final state: the -m gpu suite, smoke(), the driver's bench command and the default bench line in one box session
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/gputest.log 2>&1; echo "pytest exit $?" >> $O/gputest.log
tail -6 $O/gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2> $O/bench_driver.err; echo "bench exit $?"
tail -n 1 $O/bench_driver.log > $O/bench_driver.json; cp bench_detail.json $O/bench_driver_detail.json
timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "bench exit $?"
tail -n 1 $O/bench_default.log > $O/bench_default.json; cp bench_detail.json $O/bench_default_detail.json
wc -c $O/bench_driver.json $O/bench_default.json
python - <<'PY'
import json
for f in ("bench_driver","bench_default"):
    d=json.load(open(f"gpurun_out/r06f/{f}.json"))
    print(f, d["value"], d["roofline"]["frac"], d["legs"]["closed_loop"], d["raycast"]["kernel_us"])
PY
