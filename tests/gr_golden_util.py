"""Loading the GR fixtures of tests/golden/gen_gr_golden.py (the reference's own mma_conv.py / mask_aggr.py run on CPU over stand-ins for its
absent third-party imports) for the oracle test (CPU) and the HIP parity test (GPU)."""
import ast
import glob
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GR_FIXTURES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(HERE, "golden", "gr_*.npz")))


class GRFixture:
    def __init__(self, name):
        self.name = name
        self.d = np.load(os.path.join(HERE, "golden", name + ".npz"))
        self.cfg = ast.literal_eval(str(self.d["meta"]))
        self.T, self.F = self.cfg["towers"], self.cfg["F"]
        self.divide_input = self.cfg.get("divide_input", False)
        self.pre_layers, self.post_layers = self.cfg.get("pre_layers", 1), self.cfg.get("post_layers", 1)
        self.p, self.seed = float(self.d["p"]), int(self.d["seed"])
        self.avg_deg = {"lin": float(self.d["avg_deg_lin"]), "log": float(self.d["avg_deg_log"])}
        self.cin = self.d["x"].shape[1]
        self.cout = self.d["cot"].shape[1]

    def t(self, key, dtype=torch.float32):
        return torch.from_numpy(np.ascontiguousarray(self.d[key])).to(dtype)

    def has(self, key):
        return key in self.d.files

    def param_keys(self):
        return [k[len("param/"):] for k in self.d.files if k.startswith("param/")]

    def oracle_params(self, dtype=torch.float32, requires_grad=False):
        """The parameter dict oracle/gr_oracle.conv_forward takes (stacks as lists) + the flat {key: tensor} view of the same tensors."""
        flat = {k: self.t("param/" + k, dtype).requires_grad_(requires_grad) for k in self.param_keys()}
        T = self.T
        prm = {"pre_w": [[flat["pre_w/%d/%d" % (t, l)] for l in range(self.pre_layers)] for t in range(T)],
               "pre_b": [[flat["pre_b/%d/%d" % (t, l)] for l in range(self.pre_layers)] for t in range(T)],
               "post_w": [[flat["post_w/%d/%d" % (t, l)] for l in range(self.post_layers)] for t in range(T)],
               "post_b": [[flat["post_b/%d/%d" % (t, l)] for l in range(self.post_layers)] for t in range(T)],
               "lin_w": flat["lin_w"], "lin_b": flat["lin_b"]}
        if "enc_w" in flat:
            prm["enc_w"], prm["enc_b"] = flat["enc_w"], flat["enc_b"]
        return prm, flat
