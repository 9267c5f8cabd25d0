"""Firing-count instrumentation: the counterpart of tools/cal_firing_num.py:138-174, 203-225, 272-285.

The reference registers a forward hook on every Q_IFNode and accumulates mean(output * quant) / test_num per module
name.  Here each neuron's kernel launch adds {sum of integer counts, number of non-zero counts} into a 2-word device
counter (wave-level reduction inside s2f_lif_fwd), so recording costs no extra pass over the activations."""
import json
from collections import OrderedDict

import torch

from . import ops
from .neuron import Q_IFNode


class FiringRecorder:
    def __init__(self, model, quant=8):
        self.model, self.quant = model, quant
        self.nodes = OrderedDict((n, m) for n, m in model.named_modules() if isinstance(m, Q_IFNode))
        self.table = OrderedDict()
        self.nonzero = OrderedDict()
        self.num_images = 0

    def __enter__(self):
        dev = next(self.model.parameters()).device
        for m in self.nodes.values():
            m.stats = ops.new_stats(dev)
            m.stats_elems = 0
        return self

    def __exit__(self, *exc):
        for m in self.nodes.values():
            m.stats = None
        return False

    def collect(self):
        """Call after each forward: folds this forward's counters into the running table (one D2H copy)."""
        names = [n for n, m in self.nodes.items() if m.stats_elems > 0]
        if not names:
            return
        st = torch.stack([ops.read_stats(self.nodes[n].stats) for n in names]).cpu()
        for i, n in enumerate(names):
            m = self.nodes[n]
            rate = float(st[i, 0]) / m.stats_elems * (self.quant / m.D)     # == mean(output * quant)
            self.table[n] = self.table.get(n, 0.0) + rate
            self.nonzero[n] = self.nonzero.get(n, 0.0) + float(st[i, 1]) / m.stats_elems
            m.stats.zero_()
            m.stats_elems = 0
        self.num_images += 1

    def result(self, test_num=None):
        k = test_num or max(self.num_images, 1)
        return {"t0": OrderedDict((n, v / k) for n, v in self.table.items())}

    def to_json(self, test_num=None):
        return json.dumps(self.result(test_num))

    def to_csv(self, path, test_num=None):
        """Same shape as the reference's fr_rate.csv: index = module name, one column 'T' (cal_firing_num.py:272-285)."""
        with open(path, "w") as f:
            f.write(",T\n")
            for n, v in self.result(test_num)["t0"].items():
                f.write(f"{n},{v}\n")
