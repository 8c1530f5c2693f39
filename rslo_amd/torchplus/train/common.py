"""Run-folder helper (reference: rslo/torchplus/train/common.py:6-22)."""
import datetime
import os
import shutil
from pathlib import Path


def create_folder(prefix, add_time=True, add_str=None, delete=False):
    """Makes and returns the run directory `prefix[/<yymmdd_HHMMSS>[_<add_str>]]`.
    `delete` starts from an empty `prefix` tree (and an empty leaf, should the time stamp collide)."""
    root = Path(prefix)
    if delete:
        shutil.rmtree(root, ignore_errors=True)
        root.mkdir(parents=True)
    leaf = root
    if add_time:
        stamp = datetime.datetime.now().strftime("%y%m%d_%H%M%S")
        leaf = root / (stamp if add_str is None else "%s_%s" % (stamp, add_str))
    if delete and leaf != root:
        shutil.rmtree(leaf, ignore_errors=True)
    os.makedirs(leaf, exist_ok=(leaf == root and delete))
    return str(leaf)
