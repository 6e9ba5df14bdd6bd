#!/bin/bash
# End-to-end exercise of the training entrypoint over option combinations (small sizes); run on a GPU box:
#   gpurun -- bash tools/cli_check.sh
set -e
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/cli
rm -rf $O; mkdir -p $O
COMMON="--residual_blocks 2 --factor 2 --hr_height 32 --hr_width 32 --batch_size 4 --n_batches 6 --warmup_batches 2 --report_freq 1 --synthetic_batches 4 --dataset_type synthetic --root $O"
echo "== conditional"; python tools/train.py $COMMON --name cond --conditional true --d_channels 8 16 16 32 2>&1 | tail -1 | cut -c1-200
echo "== heads + hist"; python tools/train.py $COMMON --name heads --lambda_hist 0.05 --bins 4 --sigma 5 --lambda_hit 2.5 --lambda_mask 0.5 --lambda_nnz 1e-6 2>&1 | tail -1 | cut -c1-200
echo "== resume"; python tools/train.py $COMMON --name heads --load_checkpoint $O/saved_models/heads_generator_2.pth --lambda_hist 0.05 --bins 4 --sigma 5 2>&1 | tail -1 | cut -c1-200
echo "== json default"; echo '{"lambda_adv": 0.02, "res_scale": 0.2, "update_d": 2}' > $O/over.json; python tools/train.py $COMMON --name js --default $O/over.json 2>&1 | tail -1 | cut -c1-200
echo "== scaling_power + E_thres + final res blocks"; python tools/train.py $COMMON --name sp --scaling_power 0.5 --E_thres 0.1 --num_final_res_blocks 1 --lambda_pow 0.5 2>&1 | tail -1 | cut -c1-200
echo "== update_g 2, d_threshold high"; python tools/train.py $COMMON --name ug --update_g 2 --d_threshold 10 2>&1 | tail -1 | cut -c1-200
echo "== standard discriminator"; python tools/train.py $COMMON --name std --discriminator standard --d_channels 8 16 2>&1 | tail -1 | cut -c1-200
echo "== non-relativistic"; python tools/train.py $COMMON --name nr --relativistic false 2>&1 | tail -1 | cut -c1-200
echo "== drop_rate"; python tools/train.py $COMMON --name dr --drop_rate 0.2 2>&1 | tail -1 | cut -c1-200
echo "== transposed-conv upsampling, n_checkpoints, save_late, reference option file keys"; echo '{"n_cpu": 0, "sample_interval": -1, "validation_interval": 1000, "evaluation_interval": 1000, "n_epochs": 20}' > $O/const.json; python tools/train.py $COMMON --name tc --factor 4 --use_transposed_conv true --n_checkpoints 3 --save_late 2 --default $O/const.json 2>&1 | tail -1 | cut -c1-200; ls $O/saved_models | grep -c "tc_"
echo "== 3 channels, factor 4"; python tools/train.py --residual_blocks 1 --factor 4 --hr_height 32 --hr_width 48 --channels 3 --batch_size 2 --n_batches 4 --warmup_batches 1 --report_freq 1 --synthetic_batches 3 --dataset_type synthetic --root $O --name c3 2>&1 | tail -1 | cut -c1-200
echo "== sparse jets from a .npy row table"
python - <<'PY'
import numpy as np
rng = np.random.RandomState(0); L = 40; rows = np.zeros((16, 2 * L + 1), dtype=np.float32)
for b in range(16):
    n = rng.randint(5, L); rows[b, 0:2 * n:2] = rng.randint(0, 32 * 32, size=n); rows[b, 1:2 * n:2] = rng.rand(n) * 10 + 0.1
np.save("gpurun_out/cli/jets.npy", rows)
PY
python tools/train.py $COMMON --name jets --dataset_type spjet --dataset_path $O/jets.npy 2>&1 | tail -1 | cut -c1-200
ls $O/saved_models | wc -l; rm -rf $O
