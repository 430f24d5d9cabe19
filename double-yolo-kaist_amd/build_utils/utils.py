"""Loss, target assignment, IoU, NMS and box utilities of the reference's build_utils/utils.py on the
MI355X HIP path (same names and signatures; reference build_utils/utils.py:24-469).  Importing this
module needs neither cv2 nor torchvision."""
import glob
import math
import os
import random

import numpy as np
import torch
import torch.nn as nn


def init_seeds(seed=0):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def check_file(file):
    """return `file` if it exists, else the first recursive glob match (reference utils.py:30-37)"""
    if os.path.isfile(file):
        return file
    files = glob.glob("./**/" + file, recursive=True)
    assert len(files), "File Not Found: %s" % file
    return files[0]


def get_yolo_layers(model):
    return [i for i, d in enumerate(model.module_defs) if d["type"] == "yolo"]
