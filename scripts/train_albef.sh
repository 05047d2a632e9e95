#!/usr/bin/env bash
# Entry point mirroring the reference's src/train_albef.sh (accelerate launch ... src/train/main.py --encoder_name
# albef_no_distill --optimizer_mode dat, train_albef.sh:1-18): same flags; one process per MI355X, clients are dealt
# round-robin to the ranks, the per-round FedAvg of adapter_1 (30 modules, 8.95 MB) is one RCCL all-reduce.
NGPUS=${NGPUS:-1}
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPUS" --master-addr 127.0.0.1 --master-port "${PORT:-29512}" \
  -m feddat_amd.train \
  --encoder_name albef_no_distill --optimizer_mode dat \
  --pretrained_model_name ./models/ALBEF.pth \
  --ordered_cl_tasks domain \
  --climb_data_dir '' --do_train \
  --model_path ./models/ --output_dir ./logs/ \
  --batch_size 2 --val_batch_size 2 --lr 1e-4 --seed 2 \
  --adapter_reduction_factor 16 --adapter_config pfeiffer --splits train_small val test "$@"
