"""YAML -> attribute bag with the reference's key surface (configs/config.py:25-126).

Required keys: method task aug_list checkpoint loss_type tasks_per_batch max_ctx_num noise_scale lr
weight_decay optimizer bg_gen_freq val_iters val_freq device seed; everything else is optional
with the reference's defaults.  Like the reference, constructing from a file creates
results/<mode>/<method>/<timestamp...>/, dumps the config there and attaches a file logger.
"""
import logging
import os
from time import strftime

import torch
import yaml

_OPTIONAL = {   # attribute -> (yaml key, default)
    "mode": ("mode", "train"), "agg_mode": ("agg_mode", None), "img_agg": ("img_agg", None), "gen_bg": ("gen_bg", True),
    "output_mask": ("output_mask", False), "contrastive": ("contrastive", False), "contrastive_rate": ("contrastive_rate", 1),
    "temperature": ("temperature", 0.07), "data_size": ("data_size", None), "dim_w": ("dim_w", None),
    "n_hidden_units_r": ("n_hidden_units_r", None), "dim_r": ("dim_r", None), "dim_z": ("dim_z", None),
    "num_steps": ("num_updates", None), "test_num_steps": ("test_num_updates", None), "dim_hidden": ("num_filters", None),
    "first_order": ("first_order", None), "update_lr": ("update_lr", None), "beta": ("beta", 0), "tsne": ("tsne", False),
    "iterations": ("iterations", 50000),
}
_REQUIRED = ["method", "task", "aug_list", "checkpoint", "loss_type", "tasks_per_batch", "max_ctx_num", "noise_scale", "lr",
             "weight_decay", "optimizer", "bg_gen_freq", "val_iters", "val_freq", "seed"]
_TASK_SHAPES = {   # task -> (img_size, input_dim, output_dim)
    "shapenet_3d": ([64, 64, 4], 4, 4), "shapenet_3d_segmentation": ([64, 64, 4], 4, 4),
    "pascal_1d": ([128, 128, 1], 1, 1), "shapenet_1d": ([128, 128, 1], 3, 2), "distractor": ([128, 128, 1], 2, 2),
}


class Config(object):
    def __init__(self, config=None):
        if config:
            with open(config, "rb") as f:
                self.set_init_values(yaml.safe_load(f))

    def set_init_values(self, cfg, side_effects=True):
        for key in _REQUIRED:
            setattr(self, key, cfg[key])
        for attr, (key, default) in _OPTIONAL.items():
            setattr(self, attr, cfg.get(key, default))
        self.device = torch.device(cfg["device"])
        self.timestamp = strftime("%Y-%m-%d_%H-%M-%S")
        if self.task not in _TASK_SHAPES:
            raise TypeError(f"{self.task} is not implemented in this experiments!")
        self.img_size, self.input_dim, self.output_dim = (list(_TASK_SHAPES[self.task][0]),) + _TASK_SHAPES[self.task][1:]
        self.save_path = (f"results/{self.mode}/{self.method}/{self.timestamp}_{self.task}_datasize_{self.data_size}_"
                          f"{self.agg_mode}_{self.img_agg}{self.loss_type}_{self.aug_list}_seed_{self.seed}")
        if side_effects:
            self.create_dirs()
            self.save_config()
            self.add_logger()

    def create_dirs(self):
        os.makedirs(f"{self.save_path}/models", exist_ok=True)

    def save_config(self):
        with open(os.path.join(self.save_path, "config.yml"), "w") as f:
            yaml.dump(self.__dict__, f)

    def add_logger(self):
        logging.basicConfig(level=logging.INFO, format="%(message)s")
        self.logger = logging.getLogger()
        fh = logging.FileHandler(f"{self.save_path}/log.log", "a")
        fh.setLevel(logging.INFO)
        self.logger.addHandler(fh)
