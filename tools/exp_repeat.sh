#!/bin/bash
# run-to-run spread of the bench line:  tools/exp_repeat.sh [n] [bench args / ENV=value ...]
N=${1:-4}; shift
A="--steps 20 --warmup 5 --no-cpu-baseline --no-hbm-leg"
envs=""; args=""
for t in "$@"; do case $t in *=*) envs="$envs $t";; *) args="$args $t";; esac; done
for i in $(seq $N); do env $envs timeout 300 python3 bench.py $A $args 2> >(grep "host submit" | cut -c1-220 >&2) | python3 tools/bench_line.py; done
