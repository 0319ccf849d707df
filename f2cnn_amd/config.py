"""configF2CNN.conf reader: same file, sections and keys as the reference's configure.py:23-35."""
import os
from configparser import ConfigParser

DEFAULTS = {"FRAMERATE": 16000, "NCHANNELS": 128, "LOW_FREQ": 100, "RADIUS": 5, "SAMPLING_PERIOD": 10000}
CONFIG_NAME = "configF2CNN.conf"


class F2Config:
    def __init__(self, path=CONFIG_NAME):
        cp = ConfigParser()
        self.found = bool(cp.read(path))
        g = lambda sec, key: cp.getint(sec, key) if cp.has_option(sec, key) else DEFAULTS[key]
        self.framerate = g("FILTERBANK", "FRAMERATE")
        self.nchannels = g("FILTERBANK", "NCHANNELS")
        self.low_freq = g("FILTERBANK", "LOW_FREQ")
        self.radius = g("CNN", "RADIUS")
        self.sampling_period = g("CNN", "SAMPLING_PERIOD")

    @property
    def dots_per_input(self):
        return 2 * self.radius + 1                      # InputGenerator.py:61

    @property
    def step(self):
        return int(self.framerate * self.sampling_period / 1000000)   # InputGenerator.py:65


def write_default(path=CONFIG_NAME, **over):
    """Non-interactive equivalent of configure.py with its default answers."""
    v = {"formant": 2, "framerate": 16000, "nchannels": 128, "low_freq": 100, "sampling_period": 10000,
         "centered": True, "radius": 5, "batch_size": 32, "epochs": 20, "risk": 0.05}
    v.update(over)
    cp = ConfigParser()
    cp.add_section("FILTERBANK")
    for k in ("framerate", "nchannels", "low_freq"):
        cp["FILTERBANK"][k.upper()] = str(v[k])
    cp.add_section("CNN")
    for k in ("formant", "centered", "radius", "batch_size", "epochs", "risk", "sampling_period"):
        cp["CNN"][k.upper()] = str(v[k])
    with open(path, "w") as fp:
        cp.write(fp)
    return os.path.abspath(path)
