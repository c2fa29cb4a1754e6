"""Learning-rate multiplier of the distillation trainer: a linear ramp followed by half a cosine period down to a floor.

``LambdaWarmUpCosineScheduler`` keeps the name and constructor keywords the reference's YAML configs instantiate
(``ldm/modules/lr_scheduler.py:5-34``; ``v1-distill-arc2face-ada.yaml`` scheduler_config) because those strings are the API;
the multiplier itself is a pure function of the step, ``lr_multiplier`` below, pinned against the reference's values in
``tests/test_train_host.py`` (``train.npz``).  Use with a base learning rate of 1.0 inside ``torch.optim.lr_scheduler.LambdaLR``."""
import math


def lr_multiplier(step: int, ramp_steps: int, start: float, peak: float, floor: float, decay_end: int) -> float:
    """step < ramp_steps: the straight line from `start` (step 0) to `peak` (step ramp_steps).  Afterwards: `floor` plus a raised-cosine
    share of (peak - floor) that falls from 1 at ramp_steps to 0 at decay_end and stays there."""
    if step < ramp_steps:
        return start + (peak - start) * (step / ramp_steps)
    progress = min(1.0, (step - ramp_steps) / (decay_end - ramp_steps))
    return floor + (peak - floor) * 0.5 * (1.0 + math.cos(math.pi * progress))


class LambdaWarmUpCosineScheduler:
    """Callable step -> multiplier.  ``verbosity_interval`` is accepted for config compatibility; the trainer logs the rate itself."""

    def __init__(self, warm_up_steps, lr_min, lr_max, lr_start, max_decay_steps, verbosity_interval=0):
        self._shape = (int(warm_up_steps), float(lr_start), float(lr_max), float(lr_min), int(max_decay_steps))
        self.verbosity_interval = verbosity_interval
        self.last_lr = 0.0

    def schedule(self, n, **_):
        self.last_lr = lr_multiplier(int(n), *self._shape)
        return self.last_lr

    __call__ = schedule
