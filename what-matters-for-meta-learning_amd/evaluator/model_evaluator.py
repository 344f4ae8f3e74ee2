"""Loss-versus-context-size evaluation of a trained CNP / ANP (reference: evaluator/model_evaluator.py:95-179).

`evaluate()` sweeps the context size 1..config.max_ctx_num; per size it runs config.val_iters forward-only batches
(`model(..., test=True)` under no_grad, test-mode loss: degree error for shapenet_1d, trainer/losses.py:63-76) on the
validation split and, except for pascal_1d, on the test split, then writes `val_losses.txt` / `test_losses.txt`
(columns: context size, mean, std) and the model's state_dict, like the reference.

MI355X side: every batch runs the forward-only HIP path (one C call); the per-batch losses stay on the device and are
fetched once per sweep point (the reference syncs twice per point as well, but copies every batch synchronously);
a data source with `get_batch_u8` is read through mlhot.ingest.BatchIngest with the next batch's uint8 copy in flight
while the current forward runs.  `refine()` (model_evaluator.py:33-93) feeds `None` contexts, which the CNP / ANP
plugins of the reference do not accept either (ANPShapeNet1D.py:127) - it belongs to the SingleTask baselines and is out
of scope here.
"""
import numpy as np
import torch

from evaluator.base_evaluator import BaseEvaluator


class ModelEvaluator(BaseEvaluator):
    def __init__(self, model, loss, config, data, optimizer=None):
        super().__init__(model=model, loss=loss, config=config, optimizer=optimizer)
        self.data = data
        self.ingest = None
        if hasattr(data, "get_batch_u8") and torch.device(config.device).type == "cuda" and getattr(config, "ingest_u8", True):
            from mlhot.ingest import BatchIngest
            self.ingest = BatchIngest(config.device)

    def _log(self, msg):
        logger = getattr(self.config, "logger", None)
        if logger is not None:
            logger.info(msg)

    def refine(self):
        raise NotImplementedError("refinement drives the SingleTask baselines (out of scope, DESIGN.md §7); the CNP / ANP "
                                  "plugins take no `None` context in the reference either")

    # ------------------------------------------------------------------------------------------------------
    def _sweep(self, sources):
        """Context sizes 1..max_ctx_num, the sources interleaved per size in the reference's order (model_evaluator.py:103-110:
        its loaders may draw from numpy's global generator, so the order of the calls is part of the contract)."""
        res = {src: ([], []) for src in sources}
        for ctx_num in range(1, self.config.max_ctx_num + 1):
            for src in sources:
                loss, std = self._validate_iter(source=src, max_ctx_num=ctx_num)
                res[src][0].append(loss)
                res[src][1].append(std)
        return res

    def _save(self, name, losses, stds):
        index = list(range(1, self.config.max_ctx_num + 1))
        np.savetxt(f"{self.config.save_path}/{name}", np.column_stack((index, losses, stds)), fmt="%1.4f")

    def evaluate(self):
        self._log("\n================== Start Evaluation ===================")
        sources = ["validation"] + ([] if self.config.task == "pascal_1d" else ["test"])
        res = self._sweep(sources)
        self._save("val_losses.txt", *res["validation"])
        if "test" in res:
            self._save("test_losses.txt", *res["test"])
        torch.save(self.model.state_dict(), f"{self.config.save_path}/models/model.pt")
        self._log(f"models have been saved to {self.config.save_path}")
        self._plot(res["validation"], res.get("test"))
        return res["validation"], res.get("test")

    def evaluate_one_task(self):
        self._log("\n================== Start Evaluation ===================")
        test = self._sweep(["test"])["test"]
        self._save("test_losses.txt", *test)
        torch.save(self.model.state_dict(), f"{self.config.save_path}/models/model.pt")
        self._plot(None, test)
        return test

    # ------------------------------------------------------------------------------------------------------
    def _host_batch(self, source, shot):
        ctx_x, qry_x, ctx_y, qry_y = self.data.get_batch(source=source, tasks_per_batch=self.config.tasks_per_batch, shot=shot)
        dev = self.config.device
        return ctx_x.to(dev), qry_x.to(dev), ctx_y.to(dev), qry_y.to(dev)

    def _validate_iter(self, source, max_ctx_num=0):
        """Mean and std of the test-mode loss over config.val_iters batches with `max_ctx_num` context shots."""
        self.model.eval()
        self.data.test_counter = 0
        rng = getattr(self.data, "test_rng" if source == "test" else "val_rng", None)
        if rng is not None:
            rng.seed(42)
        n = self.config.val_iters
        vals = []
        with torch.no_grad():
            def stage():
                return self.ingest.stage(*self.data.get_batch_u8(source=source, tasks_per_batch=self.config.tasks_per_batch,
                                                                 shot=max_ctx_num))
            ticket = stage() if self.ingest is not None and n > 0 else None
            for i in range(n):
                if self.ingest is None:
                    ctx_x, qry_x, ctx_y, qry_y = self._host_batch(source, max_ctx_num)
                else:
                    ctx_x, qry_x, ctx_y, qry_y = self.ingest.take(ticket)
                if getattr(self.config, "contrastive", False):
                    pr_mu, pr_var, _, _ = self.model(ctx_x, ctx_y, qry_x, qry_y, test=True)
                else:
                    pr_mu, pr_var, _ = self.model(ctx_x, ctx_y, qry_x, test=True)
                vals.append(self.loss.calc_loss(pr_mu, pr_var, qry_y, test=True).view(1))
                if self.ingest is not None and i + 1 < n:
                    ticket = stage()                       # next batch's copy overlaps with this forward
            vals = torch.cat(vals)
            loss = vals.mean()
            std = vals.std() if vals.numel() > 1 else vals.new_full((), float("nan"))     # torch.std of one value is nan
            loss, std = loss.item(), std.item()
        self._log(f"{source} loss: {loss:.4f}")
        self._log(f"{source} std: {std:.4f}")
        return loss, std

    def _plot(self, val, test):
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:                                   # noqa: BLE001 - plotting is optional
            return
        index = np.arange(1, self.config.max_ctx_num + 1)
        for label, res in (("val", val), ("test", test)):
            if res is None:
                continue
            m, s = np.asarray(res[0]), np.asarray(res[1])
            plt.plot(index, m, label=label)
            plt.fill_between(index, m - s, m + s, alpha=0.1)
        plt.legend(loc="best")
        plt.xlabel("ctx_num")
        plt.ylabel("error")
        plt.savefig(f"{self.config.save_path}/loss_vs_ctx_num.png")
        plt.clf()
