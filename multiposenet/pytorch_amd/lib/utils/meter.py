"""Running mean / standard deviation of logged values — the reference's ``lib/utils/meter.py:16-45`` contract."""
import math

import numpy as np


class AverageValueMeter(object):
    def __init__(self):
        self.reset()

    def add(self, value, n=1):
        self.sum += value
        self.var += value * value
        self.n += n

    def value(self):
        """(mean, std): (nan, nan) when empty, (value, inf) after one sample, sample standard deviation after that."""
        if self.n == 0:
            return np.nan, np.nan
        if self.n == 1:
            return self.sum, np.inf
        mean = self.sum / self.n
        return mean, math.sqrt(max(self.var - self.n * mean * mean, 0.0) / (self.n - 1.0))

    def reset(self):
        self.sum, self.n, self.var = 0.0, 0, 0.0

    def __float__(self):
        return self.value()[0]
