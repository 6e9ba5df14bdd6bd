# final state of round 4: whole GPU suite, smoke (the profile set: tools/debug/prof_r04.sh, a call of its own)
set -o pipefail
out=gpurun_out/r04m
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
