"""A reference checkpoint's state_dict loads into elf_amd.net.PolicyValueNet through the key map of
elf_amd.net.map_reference_state_dict (the reference's parameter names differ -- init_conv(.module).N...,
resnet(.module).resnet.N.conv_lower... -- and no loader existed)."""
import os
import re

import pytest
import torch

REF = "/root/reference/src_py/elfgames/go/df_model3.py"


def reference_key_names(blocks, data_parallel):
    """The key names Model_PolicyValue.state_dict() produces, derived from the module structure in df_model3.py:20-110,167-215:
    Sequential(Conv2d, BatchNorm2d[, ReLU]) members, GoResNet.resnet = Sequential(Block...), Block.conv_lower / conv_upper."""
    mod = ".module" if data_parallel else ""
    conv = ["0.weight", "0.bias", "1.weight", "1.bias", "1.running_mean", "1.running_var", "1.num_batches_tracked"]
    keys = ["init_conv%s.%s" % (mod, c) for c in conv]
    keys += ["%s.%s" % (h, c) for h in ("pi_final_conv", "value_final_conv") for c in conv]
    keys += ["%s.%s" % (l, p) for l in ("pi_linear", "value_linear1", "value_linear2") for p in ("weight", "bias")]
    for i in range(blocks):
        keys += ["resnet%s.resnet.%d.conv_%s.%s" % (mod, i, half, c) for half in ("lower", "upper") for c in conv]
    return keys


@pytest.mark.parametrize("data_parallel", [False, True])
def test_reference_state_dict_loads(data_parallel):
    from elf_amd.net import PolicyValueNet, load_reference_checkpoint, map_reference_state_dict
    blocks, dim = 3, 16
    torch.manual_seed(1)
    mine = PolicyValueNet(9, 18, blocks, dim).eval()
    own = mine.state_dict()
    # a "reference checkpoint": the same tensors under the reference's names
    ref_keys = reference_key_names(blocks, data_parallel)
    back = {re.sub(r"^resnet\.(\d+)\.(lower|upper)\.", lambda m: "resnet%s.resnet.%s.conv_%s." % (".module" if data_parallel else "", m.group(1), m.group(2)), k)
            .replace("init_conv.", "init_conv.module." if data_parallel else "init_conv."): v for k, v in own.items()}
    assert sorted(back) == sorted(ref_keys)
    ckpt = {"state_dict": back, "step": 7, "options": {}}          # rlpytorch/model_base.py:96-100
    assert sorted(map_reference_state_dict(ckpt)) == sorted(own)
    net = load_reference_checkpoint(ckpt, board_size=9)
    s = (torch.rand(4, 18, 9, 9) < 0.3).float()
    with torch.no_grad():
        a, b = mine({"s": s}), net({"s": s})
    assert torch.equal(a["pi"], b["pi"]) and torch.equal(a["V"], b["V"])
    with pytest.raises(KeyError):
        map_reference_state_dict({"state_dict": {"mystery.weight": torch.zeros(1)}})


def test_key_names_follow_the_reference_source():
    """the structural facts reference_key_names relies on, read off the reference source where it is present"""
    if not os.path.exists(REF):
        pytest.skip("/root/reference not present")
    src = open(REF).read()
    for needle in ("self.conv_lower = self._conv_layer()", "self.conv_upper = self._conv_layer(relu=False)", "self.resnet = nn.Sequential(*self.blocks)",
                   "self.init_conv = self._conv_layer(last_planes)", "self.pi_final_conv = self._conv_layer(self.options.dim, 2, 1)",
                   "self.value_final_conv = self._conv_layer(self.options.dim, 1, 1)", "self.pi_linear = nn.Linear(d * 2, d + 1)",
                   "self.value_linear1 = nn.Linear(d, 256)", "self.value_linear2 = nn.Linear(256, 1)", "self.resnet = GoResNet(option_map, params)",
                   "nn.DataParallel(\n                    self.init_conv"):
        assert needle in src, needle
