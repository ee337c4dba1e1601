"""Model-side checkpoint format of the reference: what `Trainer.save_checkpoint` writes and
`Trainer.load_checkpoint(model_only=True)` reads (nerf/utils.py:1302-1396).

A checkpoint is either a bare state_dict or a dict with `model` (state_dict; "best" checkpoints drop
`density_grid`, :1351-1352) plus the renderer scalars `mean_count`, `mean_density`, `mean_density_torso`.
Parameter / buffer names are identical in this tree's NeRFNetwork, so reference files load unchanged.
"""
import torch


def model_state(model):
    """The model part of a reference checkpoint for `model`."""
    return {"model": model.state_dict(), "mean_count": model.mean_count, "mean_density": model.mean_density,
            "mean_density_torso": model.mean_density_torso}


def save_checkpoint(model, path, best=False, **extra):
    state = model_state(model)
    if best and "density_grid" in state["model"]:
        state["model"] = {k: v for k, v in state["model"].items() if k != "density_grid"}
    state.update(extra)
    torch.save(state, path)
    return path


def load_checkpoint(model, checkpoint, map_location=None):
    """Load a reference checkpoint (path or already-loaded dict). Returns (missing_keys, unexpected_keys)."""
    if not isinstance(checkpoint, dict):
        checkpoint = torch.load(checkpoint, map_location=map_location, weights_only=False)
    if "model" not in checkpoint:
        model.load_state_dict(checkpoint)
        return [], []
    missing, unexpected = model.load_state_dict(checkpoint["model"], strict=False)
    for key in ("mean_count", "mean_density", "mean_density_torso"):
        if key in checkpoint:
            setattr(model, key, checkpoint[key])
    if getattr(model, "enc_a", None) is not None:
        model.enc_a = None  # the lip-smoothing state belongs to a stream, not to the weights
    return list(missing), list(unexpected)
