"""Base class of the trainers (reference: trainer/base_trainer.py:23-55)."""
import os


class BaseTrainer:
    def __init__(self, model, loss, optimizer, config):
        self.config = config
        self.model = model
        self.loss = loss
        self.optimizer = optimizer
        self.iterations = config.iterations
        self.start_iter = 1
        self.best_loss = {"validation": 50000, "test": 20000}      # the reference's initial values (base_trainer.py:25)
        self.writer = None
        import torch.distributed as dist
        rank0 = not dist.is_initialized() or dist.get_rank() == 0
        if rank0:
            try:                                       # TensorBoard is optional (not in the MI355X image)
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter(config.save_path)
            except Exception:                          # noqa: BLE001
                self.writer = None
            os.makedirs(os.path.join(config.save_path, "models"), exist_ok=True)

    def train(self):
        raise NotImplementedError

    def _train_iter(self, it):
        raise NotImplementedError
