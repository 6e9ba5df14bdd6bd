# headline step with the wino24 weight gradient: stream schedules once more (weight gradients on a second stream or not; D phase beside the backward or not)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r04x
B="python3 bench.py --steps 20 --warmup 5 --no-configs --no-alt --no-cpu-baseline --no-kernel-timing"
for r in 1 2; do
  for v in "default" "SRK_OVERLAP_WGRAD=0" "SRK_D_OVERLAP=0" "SRK_WGRAD_WINO_TARGET=240" "SRK_WGRAD_W22_FORM=1"; do
    if [ "$v" = default ]; then ms=$($B 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],2))"); else ms=$(env $v $B 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],2))"); fi
    echo "round $r $v: $ms ms" | tee -a gpurun_out/r04x/schedules.txt
  done
done
