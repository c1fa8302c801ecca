#!/bin/bash
# End-to-end runs of the three drop-in entry points on synthetic data (a few steps each): argument parsing, config merge,
# logger, data loaders, training loop, evaluation, checkpoint write.  Needs a GPU.  Usage: bash tools/cli_smoke.sh [outdir]
set -e
OUT=${1:-/tmp/lr2ppo_cli_smoke}   # 8 GB of checkpoints: keep them out of gpurun_out/
mkdir -p "$OUT"
COMMON="--config_path lr2ppo_amd/configs/roberta_base.json --vit_config_path lr2ppo_amd/configs/vit_base_16_224.json \
 --train_path none --dev_path none --seq_length 196 --max_imgs 16 --visual_feat_dim 768 --learning_rate 1e-4 --batch_size 2"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29671
echo "== stage 1 (pointwise)"
python -m lr2ppo_amd.finetune.pointwise $COMMON --mode reg --max_tags 20 --epochs_num 1 --report_steps 2 --synthetic_items 8 \
  --synthetic_val_items 3 --max_steps 2 --output_model_path "$OUT/stage1.bin" --log_path "$OUT/stage1.log"
echo "== stage 2 (pairwise reward)"
python -m lr2ppo_amd.finetune.reward_pair_dataloader $COMMON --mode cls --epochs_num 1 --report_steps 2 --synthetic_items 8 \
  --synthetic_val_items 4 --max_steps 2 --output_model_path "$OUT/stage2.bin" --log_path "$OUT/stage2.log"
echo "== stage 3 (PPO) from the stage-1 / stage-2 checkpoints"
python -m lr2ppo_amd.finetune.ppo $COMMON --mode reg --epochs_num 2 --critic_learning_rate 1e-4 --max_timesteps 1 \
  --update_timesteps 2 --kl_div_loss_weight 0.001 --entropy_weight 0.001 --value_clip 0.5 --synthetic_items 4 \
  --synthetic_val_items 3 --max_cycles 1 --pretrained_model_path "$OUT/stage1.bin" --reward_model_path "$OUT/stage2.bin" \
  --output_model_path "$OUT/stage3.bin" --log_path "$OUT/stage3.log"
echo "== stage 3 at sequence length 1 (ppo_trad twin), synthetic LETOR-shaped queries"
python -m lr2ppo_amd.finetune.ppo_trad $COMMON --mode reg --epochs_num 2 --critic_learning_rate 1e-4 --max_timesteps 1 \
  --update_timesteps 2 --kl_div_loss_weight 0.001 --entropy_weight 0.001 --value_clip 0.5 --synthetic_items 8 \
  --synthetic_val_items 3 --max_cycles 1 --output_model_path "$OUT/stage3_trad.bin" --log_path "$OUT/stage3_trad.log"
echo "== BASELINE configs[0]: the _trad pointwise / pairwise twins (sequence length 1), synthetic LETOR-shaped queries"
python -m lr2ppo_amd.finetune.pointwise_trad $COMMON --mode reg --epochs_num 1 --report_steps 2 --synthetic_items 8 \
  --synthetic_val_items 3 --max_steps 2 --output_model_path "$OUT/trad1.bin" --log_path "$OUT/trad1.log"
python -m lr2ppo_amd.finetune.pointwise_2data_trad $COMMON --mode reg --epochs_num 1 --report_steps 2 --synthetic_items 8 \
  --synthetic_val_items 3 --max_steps 4 --output_model_path "$OUT/trad2.bin" --log_path "$OUT/trad2.log"
python -m lr2ppo_amd.finetune.reward_trad $COMMON --mode reg --epochs_num 1 --report_steps 2 --synthetic_items 8 \
  --synthetic_val_items 4 --max_steps 2 --output_model_path "$OUT/trad_reward.bin" --log_path "$OUT/trad_reward.log"
echo "== stage 1 on RAW inputs: ViT + RoBERTa stacks (2 layers each here) in front of the head, trained end to end"
python -m lr2ppo_amd.finetune.pointwise $COMMON --mode reg --max_tags 2 --epochs_num 1 --report_steps 2 --synthetic_items 4 \
  --synthetic_val_items 2 --max_steps 2 --raw_inputs --finetune_encoders --encoder_layers 2 \
  --output_model_path "$OUT/stage1_raw.bin" --log_path "$OUT/stage1_raw.log"
ls -la "$OUT"
echo CLI_SMOKE_OK
