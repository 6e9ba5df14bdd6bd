# whole GPU suite with every CU's LDS poisoned (NaN patterns) in front of every conv / weight-gradient launch; then the plain suite + smoke on the final build
set -o pipefail
out=gpurun_out/r04w
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SRK_POISON_LDS=1 timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $out/pytest_poison.log 2>&1; echo "poisoned suite rc=$?"; tail -15 $out/pytest_poison.log | cut -c1-250
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
