"""Dynamic lookups, mirroring utils/program_utils.py:4-5 (only the model lookup is on the hot path)."""
import importlib


def get_model(model):
    return importlib.import_module("cycle_diffusion_amd.model.{}".format(model)).Model
