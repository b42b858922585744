"""Optimiser / schedule construction around the model, as the reference's training loop does it (src/utils.py:25-86 ``set_optim``,
called from main.py:92): parameter groups by NAME -- for ``model_name == "SNAG"`` anything containing ``weight_raw`` / ``bias`` /
``multi_loss_layer`` is not decayed and ``multi_loss_layer`` runs at 5x the learning rate -- AdamW or Adam, and the fixed / linear /
cosine schedules with warm-up.  Plain host code on torch.optim: the drop-in ``snag_amd.mmea.SNAG`` keeps the reference's parameter
names, so the groups come out identical (golden G14)."""
import torch
from torch import optim


def set_optim(opt, model_list, freeze_part=(), no_freeze_part=(), accumulation_step=None):
    """-> (optimizer, scheduler).  ``opt`` carries model_name, lr, weight_decay, adam_epsilon, optim, scheduler, warmup_steps,
    total_steps, accumulation_steps -- the fields the reference reads."""
    named = []
    for model in model_list:
        for n, p in model.named_parameters():
            trainable = not any(nd in n for nd in freeze_part) and (len(no_freeze_part) == 0 or any(nd in n for nd in no_freeze_part))
            if trainable:
                named.append((n, p))
            else:
                p.requires_grad = False
    if opt.model_name == "SNAG":
        no_decay, large_lr = ["weight_raw", "bias", "multi_loss_layer"], ["multi_loss_layer"]
        plain = lambda n: any(nd in n for nd in no_decay)
        large = lambda n: any(nd in n for nd in large_lr)
        groups = [
            {"params": [p for n, p in named if not plain(n)], "lr": opt.lr, "weight_decay": opt.weight_decay},
            {"params": [p for n, p in named if plain(n) and not large(n)], "lr": opt.lr, "weight_decay": 0.0},
            {"params": [p for n, p in named if plain(n) and large(n)], "lr": 5 * opt.lr, "weight_decay": 0.0},
        ]
    else:
        groups = [{"params": [p for _, p in named], "lr": opt.lr, "weight_decay": opt.weight_decay}]
    if opt.optim == "adamw":
        optimizer = optim.AdamW(groups, lr=opt.lr, eps=opt.adam_epsilon)
    elif opt.optim == "adam":
        optimizer = optim.Adam(groups, lr=opt.lr)
    else:
        raise ValueError(f"unknown optimiser {opt.optim!r}")
    if accumulation_step is None:
        accumulation_step = opt.accumulation_steps
    if opt.scheduler == "fixed":
        scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda step: 1.0)
    elif opt.scheduler in ("linear", "cos"):
        from transformers import get_cosine_schedule_with_warmup, get_linear_schedule_with_warmup
        make = get_linear_schedule_with_warmup if opt.scheduler == "linear" else get_cosine_schedule_with_warmup
        scheduler = make(optimizer, num_warmup_steps=int(opt.warmup_steps / accumulation_step),
                         num_training_steps=int(opt.total_steps / accumulation_step))
    else:
        raise ValueError(f"unknown scheduler {opt.scheduler!r}")
    return optimizer, scheduler
