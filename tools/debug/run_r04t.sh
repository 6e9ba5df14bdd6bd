# wino24 weight gradient: is the fetch cost latency (max over pieces) or volume?  timing-only: only dy / only x pieces fetch
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/debug/run_wgvar.sh w24f2 w24deadx w24deaddy w24dead w24f2 w24deadx w24deaddy w24dead
