"""Reference ``ldm/modules/lr_scheduler.py:5-34``: warm-up then cosine decay multiplier (use with a base lr of 1.0
inside torch.optim.lr_scheduler.LambdaLR)."""
import numpy as np


class LambdaWarmUpCosineScheduler:
    def __init__(self, warm_up_steps, lr_min, lr_max, lr_start, max_decay_steps, verbosity_interval=0):
        self.lr_warm_up_steps = warm_up_steps
        self.lr_start = lr_start
        self.lr_min = lr_min
        self.lr_max = lr_max
        self.lr_max_decay_steps = max_decay_steps
        self.last_lr = 0.0
        self.verbosity_interval = verbosity_interval

    def schedule(self, n, **kwargs):
        if self.verbosity_interval > 0 and n % self.verbosity_interval == 0:
            print(f"current step: {n}, recent lr-multiplier: {self.last_lr}")
        if n < self.lr_warm_up_steps:
            lr = (self.lr_max - self.lr_start) / self.lr_warm_up_steps * n + self.lr_start
        else:
            t = min((n - self.lr_warm_up_steps) / (self.lr_max_decay_steps - self.lr_warm_up_steps), 1.0)
            lr = self.lr_min + 0.5 * (self.lr_max - self.lr_min) * (1 + np.cos(t * np.pi))
        self.last_lr = lr
        return lr

    def __call__(self, n, **kwargs):
        return self.schedule(n, **kwargs)
