"""Host mirror of the reference's training driver around the engine:
YoloBaseTaskModel.Train (:116-215), TrainEpoch (:291-356) with its warm-up, LambdaLR / OneCycle / LrLambda / Interp
(:492-536) and the lr0 fit (:142).  Data loading / augmentation stay outside (out of scope); `fit` consumes ready batches."""
import math
import os

import numpy as np

from . import weights_bin
from .detector import Detector
from .model import AMPWrapper, v8DetectionLoss, v8SegmentationLoss


def lr_fit(nc):
    """YoloBaseTaskModel.cs:142: lr0 fit equation to 6 decimal places."""
    return round(0.002 * 5 / (4 + nc), 6)


def one_cycle(y1, y2, steps):
    """:492-502 (cosine from y1 to y2 over `steps` epochs)."""
    return lambda x: max((1 - math.cos(x * math.pi / steps)) / 2, 0) * (y2 - y1) + y1


def lr_lambda(y1, y2, steps):
    """:504-512 (linear from y1 to y2)."""
    return lambda epoch: max(1 - epoch / steps, 0) * (y1 - y2) + y2


def interp(x, xp, fp):
    """:514-536 (scalar np.interp with end clamping and exact-hit lookup)."""
    if len(xp) != len(fp) or not len(xp):
        raise ValueError("xp and fp must be non-empty and of equal length")
    if x <= xp[0]:
        return fp[0]
    if x >= xp[-1]:
        return fp[-1]
    import bisect
    i = bisect.bisect_left(xp, x)
    if xp[i] == x:
        return fp[i]
    t = (x - xp[i - 1]) / (xp[i] - xp[i - 1])
    return fp[i - 1] + t * (fp[i] - fp[i - 1])


class LrSchedule:
    """The optimizer's three ParamGroups[i].LearningRate as the reference moves them: LambdaLR stepped once per epoch
    (:160,183) and, while ni = i + nb*epoch <= nw, overwritten per iteration by the warm-up interpolation (:303-319)
    (group 0 = bias starts at WarmUpBiasLr, the others at 0; target = initial_lr * lambda(epoch), epoch 1-based)."""

    def __init__(self, nc, epochs, nb, lrf=0.01, warmup_epochs=3, warmup_bias_lr=0.1, use_cos_lr=False, lr0=None):
        self.initial = lr_fit(nc) if lr0 is None else lr0
        self.lam = one_cycle(1.0, lrf, epochs) if use_cos_lr else lr_lambda(1.0, lrf, epochs)
        self.nb, self.nw = nb, max(warmup_epochs * nb, 100)                       # :167-168
        self.warmup_bias_lr = warmup_bias_lr
        self.steps = 0
        self.lrs = [self.initial * self.lam(0)] * 3                               # LambdaLR constructor: lambda(0)

    def begin_iteration(self, epoch, i):
        ni = i + self.nb * epoch
        if ni <= self.nw:
            d = self.initial * self.lam(epoch)
            self.lrs = [interp(ni, [0, self.nw], [self.warmup_bias_lr if g == 0 else 0.0, d]) for g in range(3)]
        return list(self.lrs)

    def end_epoch(self):                                                          # lr_scheduler.step()
        self.steps += 1
        self.lrs = [self.initial * self.lam(self.steps)] * 3
        return list(self.lrs)


class Trainer:
    """Train loop of YoloBaseTaskModel.Train over the engine: per epoch TrainEpoch (warm-up + AMPWrapper.TrainStep), scheduler
    step, Val, best.bin / last.bin through the `.bin` container (fitness = -sum(val loss items), :185-196)."""

    def __init__(self, model, epochs, nb, out_dir=None, segment=False, **sched):
        self.model, self.epochs, self.out_dir = model, epochs, out_dir
        self.amp = AMPWrapper(model)
        # criterion and validator of the model's task (YoloBaseTaskModel's subclasses Detector / Segmenter / Obber / PoseDetector)
        from . import detector as D
        from .model import v8OBBLoss, v8PoseLoss
        task = 1 if segment else getattr(model, "TASK", 0)
        self.crit = {0: v8DetectionLoss, 1: v8SegmentationLoss, 2: v8OBBLoss, 3: v8PoseLoss}[task](model)
        self.validator = {0: D.Detector, 1: D.Segmenter, 2: D.Obber, 3: D.PoseDetector}[task]
        self.sched = LrSchedule(model.nc, epochs, nb, **sched)
        self.best_fitness = -float("inf")

    def train_epoch(self, batches, epoch):
        """TrainEpoch (:291-356): returns the SUM of the per-step loss items like the reference (zeros when no step ran).  The
        warm-up index i only advances on batches that trained: the reference's `continue` on an empty batch skips its i++."""
        items_sum, i = None, 0
        for data in batches:
            self.amp.lrs = self.sched.begin_iteration(epoch, i)
            if np.asarray(data["batch_idx"]).size < 1:                            # :322-325
                continue
            _, items = self.amp.TrainStep(np.ascontiguousarray(data["images"], np.float32), data, self.crit)
            items_sum = items if items_sum is None else items_sum + items
            i += 1
        self.steps_run = i
        n_items = getattr(self.crit, "N_ITEMS", 3)
        return items_sum if items_sum is not None else np.zeros(n_items, np.float32)

    def fit(self, train_batches, val_batches=None):
        """train_batches / val_batches: callables returning an iterable of batches for an epoch.  Returns the history."""
        hist = []
        for epoch in range(1, self.epochs + 1):
            tr = self.train_epoch(train_batches(), epoch)
            self.sched.end_epoch()
            rec = {"epoch": epoch, "train_loss": tr, "lr": list(self.sched.lrs)}
            if val_batches is not None:
                vloss, metrics, *more = self.validator(self.model).Val(val_batches())   # Segmenter / PoseDetector add mask / pose metrics
                rec.update(val_loss=vloss, metrics=metrics)
                if more:
                    rec.update(metrics2=more[0])
                fitness = -float(np.sum(vloss))
                if self.out_dir:
                    os.makedirs(os.path.join(self.out_dir, "weights"), exist_ok=True)
                    if fitness > self.best_fitness:
                        weights_bin.save_from(self.model, os.path.join(self.out_dir, "weights", "best.bin"))
                    weights_bin.save_from(self.model, os.path.join(self.out_dir, "weights", "last.bin"))
                self.best_fitness = max(self.best_fitness, fitness)
            hist.append(rec)
        return hist
