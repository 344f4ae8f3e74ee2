"""Base class of the evaluators (reference: evaluator/base_evaluator.py:22-64)."""
import os


class BaseEvaluator:
    def __init__(self, model, loss, config, optimizer=None):
        if config.save_path is None:
            raise ValueError("config.save_path is required")
        self.best_loss = {"validation": 10000, "test": 10000}
        self.config, self.model, self.loss, self.optimizer = config, model, loss, optimizer
        self.start_iter, self.iterations = 1, getattr(config, "iterations", 0)
        self.save_path = config.save_path
        os.makedirs(os.path.join(self.save_path, "models"), exist_ok=True)
        self.writer = None
        try:                                           # TensorBoard is optional (not in the MI355X image)
            from torch.utils.tensorboard import SummaryWriter
            self.writer = SummaryWriter(self.save_path, max_queue=10)
        except Exception:                              # noqa: BLE001
            self.writer = None

    def evaluate(self):
        raise NotImplementedError
