"""Checkpoint writer (tencentpretrain/model_saver.py:4-11): a plain state_dict, `.module` unwrapped."""
import torch


def save_model(model, model_path):
    target = model.module if hasattr(model, "module") else model
    torch.save(target.state_dict(), model_path)
