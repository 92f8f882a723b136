"""``PreTrainedModel``: checkpoint loading with the reference's key handling.

Host-side Python, mirrors /root/reference/vilbert/utils.py:703-1032 for the parts the model path uses:
``from_pretrained`` (local files / directories only - this build has no network cache), the
``gamma``/``beta`` -> ``weight``/``bias`` key renames (:945-958), base-model prefix handling
(:986-998), weight re-tying, ``.eval()`` on return (:1022) and *returning None* when the checkpoint
file does not exist (:904-923). The logging / S3 / TensorBoard helpers of the reference's utils.py are
out of scope.
"""
import logging
import os

import torch
from torch import nn

logger = logging.getLogger(__name__)

WEIGHTS_NAME = "pytorch_model.bin"
CONFIG_NAME = "config.json"


class PreTrainedModel(nn.Module):
    config_class = None
    pretrained_model_archive_map = {}
    base_model_prefix = ""

    def __init__(self, config, *inputs, **kwargs):
        super(PreTrainedModel, self).__init__()
        self.config = config

    def _tie_or_clone_weights(self, first_module, second_module):
        first_module.weight = second_module.weight

    def save_pretrained(self, save_directory):
        assert os.path.isdir(save_directory), "Saving path should be a directory"
        model_to_save = self.module if hasattr(self, "module") else self
        with open(os.path.join(save_directory, CONFIG_NAME), "w", encoding="utf-8") as f:
            f.write(model_to_save.config.to_json_string())
        torch.save(model_to_save.state_dict(), os.path.join(save_directory, WEIGHTS_NAME))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        config = kwargs.pop("config", None)
        state_dict = kwargs.pop("state_dict", None)
        kwargs.pop("cache_dir", None)
        from_tf = kwargs.pop("from_tf", False)
        output_loading_info = kwargs.pop("output_loading_info", False)
        default_gpu = kwargs.pop("default_gpu", True)
        if from_tf:
            raise NotImplementedError("TensorFlow checkpoints are not supported by this build")
        if config is None:
            raise ValueError("from_pretrained needs config=BertConfig(...)")

        if pretrained_model_name_or_path in cls.pretrained_model_archive_map:
            archive_file = cls.pretrained_model_archive_map[pretrained_model_name_or_path]
        elif os.path.isdir(pretrained_model_name_or_path):
            archive_file = os.path.join(pretrained_model_name_or_path, WEIGHTS_NAME)
        else:
            archive_file = pretrained_model_name_or_path
        if state_dict is None and not os.path.isfile(archive_file):
            # the reference logs and returns None here instead of raising
            logger.error("Model name '%s' was not found; assumed '%s' was a path but no file is there.",
                         pretrained_model_name_or_path, archive_file)
            return None
        if default_gpu:
            logger.info("loading weights file %s", archive_file)

        model = cls(config, *model_args, **kwargs)
        if state_dict is None:
            state_dict = torch.load(archive_file, map_location="cpu")

        renamed = {}
        for key in list(state_dict.keys()):
            new_key = key
            if "gamma" in new_key:
                new_key = new_key.replace("gamma", "weight")
            if "beta" in new_key:
                new_key = new_key.replace("beta", "bias")
            renamed[new_key] = state_dict[key]
        metadata = getattr(state_dict, "_metadata", None)
        state_dict = renamed

        missing_keys, unexpected_keys, error_msgs = [], [], []

        def load(module, prefix=""):
            local_metadata = {} if metadata is None else metadata.get(prefix[:-1], {})
            module._load_from_state_dict(state_dict, prefix, local_metadata, True, missing_keys,
                                         unexpected_keys, error_msgs)
            for name, child in module._modules.items():
                if child is not None:
                    load(child, prefix + name + ".")

        start_prefix, model_to_load = "", model
        has_prefix = any(s.startswith(cls.base_model_prefix) for s in state_dict.keys())
        if not hasattr(model, cls.base_model_prefix) and has_prefix:
            start_prefix = cls.base_model_prefix + "."
        if hasattr(model, cls.base_model_prefix) and not has_prefix:
            model_to_load = getattr(model, cls.base_model_prefix)
        load(model_to_load, prefix=start_prefix)

        if missing_keys and default_gpu:
            logger.info("Weights of %s not initialized from pretrained model: %s", model.__class__.__name__,
                        missing_keys)
        if unexpected_keys and default_gpu:
            logger.info("Weights from pretrained model not used in %s: %s", model.__class__.__name__,
                        unexpected_keys)
        if error_msgs and default_gpu:
            raise RuntimeError("Error(s) in loading state_dict for {}:\n\t{}".format(
                model.__class__.__name__, "\n\t".join(error_msgs)))
        if hasattr(model, "tie_weights"):
            model.tie_weights()
        model.eval()
        if output_loading_info:
            return model, {"missing_keys": missing_keys, "unexpected_keys": unexpected_keys,
                           "error_msgs": error_msgs}
        return model


# --- everything else of the reference's vilbert/utils.py (tbLogger :151-482, MultiTaskStopOnPlateau :39-148,
# cached_path / S3 helpers, ...) is logging / control side and is NOT rebuilt: unknown attributes are looked up in the
# reference's own utils.py when a reference checkout is attached (vilbert/__init__.py: attach_reference), so
# `import vilbert.utils as utils; utils.tbLogger(...)` (train_concap.py:29,346-354) keeps working against this package.
def _reference_utils():
    try:
        from . import _reference_utils as ref
    except ImportError:
        return None
    return ref


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    ref = _reference_utils()
    if ref is not None and hasattr(ref, name):
        return getattr(ref, name)
    raise AttributeError("module 'vilbert.utils' has no attribute %r%s" % (
        name, "" if ref is not None else " (no reference checkout attached: set VILBERT_REFERENCE_ROOT to use the "
        "reference's logging / caching helpers)"))
