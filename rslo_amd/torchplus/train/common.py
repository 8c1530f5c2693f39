"""Run-folder helper (reference: rslo/torchplus/train/common.py:6-22)."""
import datetime
import os
import shutil


def create_folder(prefix, add_time=True, add_str=None, delete=False):
    """Creates `prefix[/<yymmdd_HHMMSS>[_<add_str>]]`; `delete` wipes an existing tree first."""
    if delete and os.path.exists(prefix):
        shutil.rmtree(prefix)
    if delete:
        os.makedirs(prefix)
    folder = prefix
    if add_time:
        stamp = datetime.datetime.now().strftime("%y%m%d_%H%M%S")
        folder = os.path.join(prefix, stamp if add_str is None else "%s_%s" % (stamp, add_str))
    if delete and os.path.exists(folder):
        shutil.rmtree(folder)
    os.makedirs(folder)
    return folder
