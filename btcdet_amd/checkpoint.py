"""Checkpoint save / resume in the reference's format:

  * ``checkpoint_state_mult_opt`` / ``save_checkpoint``  -- /root/reference/tools/train_utils/train_utils.py:272-317:
    ``{'epoch', 'it', 'model_state', 'optimizer_state_lst', 'version'}`` written with torch.save to ``<name>.pth``;
  * ``load_params_from_file`` / ``load_params_with_optimizer_lst`` -- Detector3DTemplate's loaders
    (btcdet/models/detectors/detector3d_template.py:594-618, 650-678): partial load by key + shape (+ prefix), and full resume
    returning ``(it, epoch)``.
The model's state_dict keys equal the reference's (tests/test_reference_dropin_cpu.py) and the optimizer states use torch.optim
.Adam's layout with the reference's parameter numbering (train_step.GroupOptimizer.state_dict_lst), so checkpoints are
interchangeable for the modules of the hot path."""
import os

import torch

VERSION = "btcdet_amd+0.2"


def _model_of(model):
    return model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model


def checkpoint_state_mult_opt(model=None, optimizer_lst=None, epoch=None, it=None):
    states = None
    if optimizer_lst is not None:
        states = []
        for opt in optimizer_lst:
            if opt is None:
                states.append(None)
            elif hasattr(opt, "state_dict_lst"):       # one GroupOptimizer = the reference's optimizers, in order
                states.extend(opt.state_dict_lst())
            else:
                states.append(opt.state_dict())
    model_state = None
    if model is not None:
        sd = _model_of(model).state_dict()
        model_state = type(sd)((k, v.cpu()) for k, v in sd.items()) if model is not _model_of(model) else sd
    return {"epoch": epoch, "it": it, "model_state": model_state, "optimizer_state_lst": states, "version": VERSION}


def save_checkpoint(state, filename="checkpoint"):
    filename = "{}.pth".format(filename)
    torch.save(state, filename)
    return filename


def load_params_from_file(model, filename, to_cpu=False, prefix="", logger=None):
    """copies every checkpoint entry whose key exists with the same shape (and starts with `prefix`); returns the keys NOT updated"""
    if not os.path.isfile(filename):
        raise FileNotFoundError(filename)
    ckpt = torch.load(filename, map_location=torch.device("cpu") if to_cpu else None, weights_only=False)
    disk = ckpt["model_state"]
    own = model.state_dict()
    take = {k: v for k, v in disk.items() if k in own and own[k].shape == v.shape and k.startswith(prefix)}
    own.update(take)
    model.load_state_dict(own)
    missing = [k for k in own if k not in take]
    if logger is not None:
        logger.info("==> Done (loaded %d/%d)" % (len(take), len(own)))
    return missing


def load_params_with_optimizer_lst(model, filename, to_cpu=False, optimizer_lst=None, logger=None):
    if not os.path.isfile(filename):
        raise FileNotFoundError(filename)
    ckpt = torch.load(filename, map_location=torch.device("cpu") if to_cpu else None, weights_only=False)
    epoch, it = ckpt.get("epoch", -1), ckpt.get("it", 0.0)
    _model_of(model).load_state_dict(ckpt["model_state"])
    states = ckpt.get("optimizer_state_lst")
    if optimizer_lst is not None and states is not None:
        pos = 0
        for opt in optimizer_lst:
            if hasattr(opt, "load_state_dict_lst"):
                n = len(opt.groups)
                opt.load_state_dict_lst(states[pos:pos + n], iteration=it)
                pos += n
            else:
                opt.load_state_dict(states[pos])
                pos += 1
    return it, epoch
