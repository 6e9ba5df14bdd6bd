#!/bin/bash
# position of the transform gap in a group's second phase (W42_XA: slot 6 + XA), dense block at batch 32
for v in base xa0 xa1 xa3 xa4 xa5 base xa0 xa1 xa3 xa4 xa5; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_$v.so; fi
  echo "== $v: $(FMT=6 REPS=2 timeout -k 10 200 python tools/debug/chain_check.py 2>&1 | grep 'block at' | sed 's/.*chain \([0-9.]*\) us.*/\1/' | tr '\n' ' ')"
done
