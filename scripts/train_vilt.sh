#!/usr/bin/env bash
# Entry point mirroring the reference's src/train_vilt.sh (accelerate launch ... src/train/main.py, train_vilt.sh:1-20):
# same flags; one process per MI355X, client k runs on rank k (more clients than ranks: dealt by longest-processing-time), the per-round FedAvg of adapter_1 is
# one RCCL all-reduce.  NGPUS=1 runs all clients on one GPU exactly like the reference's sequential loop.
# --pretrained_model_name must be a LOCAL HuggingFace directory (the reference's script: ./models/vilt-b32-mlm); a missing
# path is an error -- drop the flag to run on random weights of the real architecture (synthetic benchmarks).
NGPUS=${NGPUS:-1}
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPUS" --master-addr 127.0.0.1 --master-port "${PORT:-29511}" \
  -m feddat_amd.train \
  --encoder_name vilt --optimizer_mode dat \
  --pretrained_model_name "${VILT_CKPT:-./models/vilt-b32-mlm}" \
  --ordered_cl_tasks domain \
  --climb_data_dir ./data --do_train \
  --output_dir ./outputs/vilt_dat \
  --batch_size 32 --lr 1e-4 --comm_rounds 30 --local_epochs 1 "$@"
