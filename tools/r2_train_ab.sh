#!/bin/bash
# same-seed training A/B on the synthetic speakers: bf16 storage vs fp32 storage (exact-parity mode) vs the log-mel variant
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O $R/logs $R/models; cd $R
COMMON="--synthetic --filters 32 --embedding-dimension 64 --batchsize 32 --epochs 8 --steps-per-epoch 120 --validation-steps 25 --num-evaluation-tasks 300 --workers 8"
for dt in bf16 f32; do
  timeout 900 python -m experiments.train_siamese $COMMON --dtype $dt > $O/train_$dt.log 2>&1; echo "train $dt rc=$?"
  cp logs/siamese__filters_32__embed_64__drop_0.0__pad=True.csv $O/r02_synthetic_training_history_$dt.csv
done
timeout 900 python -m experiments.train_siamese $COMMON --dtype bf16 --frontend logmel > $O/train_logmel.log 2>&1; echo "train logmel rc=$?"
cp logs/logmel_siamese__filters_32__embed_64__drop_0.0__pad=True.csv $O/r02_synthetic_training_history_logmel_bf16.csv
tail -3 $O/train_bf16.log $O/train_f32.log $O/train_logmel.log
