"""Small torch helpers the reference keeps in build_utils/torch_utils.py (only what the hot path
and the unchanged CLI scripts import: init_seeds, select_device, time_synchronized, model_info)."""
import time

import torch


def init_seeds(seed=0):
    torch.manual_seed(seed)


def select_device(device="cuda:0"):
    """device string -> torch.device, single GPU or CPU (reference torch_utils.py:35-50)"""
    cpu_request = device.lower() == "cpu"
    if device and not cpu_request:
        assert torch.cuda.is_available(), "CUDA unavailable, invalid device %s requested" % device
    device = torch.device(device)
    if not cpu_request and torch.cuda.is_available():
        di = 0 if device.index is None else device.index
        dp = torch.cuda.get_device_properties(di)
        print("Using torch %s CUDA:%d (%s, %dMB)" % (torch.__version__, di, dp.name, dp.total_memory / 1024 ** 2))
    else:
        print("Using torch %s CPU" % torch.__version__)
    return device


def time_synchronized():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()


def model_info(model, verbose=False):
    """parameter / gradient count summary (reference torch_utils.py:53-75)"""
    n_p = sum(x.numel() for x in model.parameters())
    n_g = sum(x.numel() for x in model.parameters() if x.requires_grad)
    if verbose:
        print("%5s %40s %9s %12s %20s" % ("layer", "name", "gradient", "parameters", "shape"))
        for i, (name, p) in enumerate(model.named_parameters()):
            print("%5g %40s %9s %12g %20s" % (i, name.replace("module_list.", ""), p.requires_grad, p.numel(), list(p.shape)))
    print("Model Summary: %g layers, %g parameters, %g gradients" % (len(list(model.parameters())), n_p, n_g))
