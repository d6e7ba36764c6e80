#!/bin/bash
# quick A/B of the headline step time: args = env assignments per arm, separated by '--'
run() { env "$@" python bench.py --no-cpu-baseline --no-pmc --no-companion --no-configs --no-roofline --no-dp-form --steps 200 --warmup 20 --blocks 3 --min-block-s 0.3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.4f ms  %.0f samples/s' % (d['ms_per_step'], d['value']))"; }
arm=()
for a in "$@" --; do
    if [ "$a" == "--" ]; then
        echo "== ${arm[*]}"
        run "${arm[@]}"
        arm=()
    else
        arm+=("$a")
    fi
done
