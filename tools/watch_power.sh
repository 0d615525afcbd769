#!/bin/bash
# Poll rocm-smi (power, sclk) while a command runs: is the forward power-capped?  usage: tools/watch_power.sh <out.txt> <cmd...>
out=$1; shift
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (edge|junction)" | tr '\n' ' ' ; echo; sleep 0.2; done ) > $out &
poll=$!
"$@"
kill $poll
