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


def load_checkpoint(model, checkpoint, map_location=None, optimizer=None, half_tables=False):
    """Load a reference checkpoint (path or already-loaded dict).  Returns (missing_keys, unexpected_keys), as
    `load_state_dict(strict=False)` reports them for the model part (nerf/utils.py:1376-1396).

    optimizer: restored from the checkpoint's `optimizer` entry when it has one (full checkpoints, :1320-1325); a state that
    does not fit (another parameter grouping) is skipped, as the reference does (:1408-1413).
    half_tables: build the persistent fp16 copies of the three grid tables now (GridEncoder.half_table) and make the fused
    inference engine read them (model.opt.half_tables) -- what the reference's `-O` mode re-creates by casting every table on
    every call (gridencoder/grid.py:43-44).  The copies follow the parameters: they are re-cast when a table's version changes.
    `model.checkpoint_meta` receives epoch / global_step / stats when the file has them."""
    if not isinstance(checkpoint, dict):
        checkpoint = torch.load(checkpoint, map_location=map_location, weights_only=False)
    if "model" not in checkpoint:
        model.load_state_dict(checkpoint)
        missing, unexpected = [], []
    else:
        missing, unexpected = model.load_state_dict(checkpoint["model"], strict=False)
        for key in ("mean_count", "mean_density", "mean_density_torso"):
            if key in checkpoint:
                setattr(model, key, checkpoint[key])
        model.checkpoint_meta = {k: checkpoint[k] for k in ("epoch", "global_step", "stats") if k in checkpoint}
        if optimizer is not None and "optimizer" in checkpoint:
            try:
                optimizer.load_state_dict(checkpoint["optimizer"])
                model.checkpoint_meta["optimizer_loaded"] = True
            except (ValueError, KeyError, RuntimeError):
                model.checkpoint_meta["optimizer_loaded"] = False
    if getattr(model, "enc_a", None) is not None:
        model.enc_a = None  # the lip-smoothing state belongs to a stream, not to the weights
    if half_tables:
        if getattr(model, "opt", None) is not None:
            model.opt.half_tables = True          # the fused engine then reads the fp16 copies (radnerf/fused.py: FusedState.refresh)
        for name in ("encoder", "encoder_ambient", "torso_encoder"):
            enc = getattr(model, name, None)
            if enc is not None and hasattr(enc, "half_table"):
                enc.half_table()
    return list(missing), list(unexpected)
