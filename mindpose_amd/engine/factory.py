"""``create_inferencer`` / ``create_evaluator`` with the reference's call signatures (mindpose/engine/factory.py:12-96): the
method-specific config and the dataset config are merged (dataset keys win, a warning names duplicates) and handed to the
registered engine class."""
import logging

from ..register import entrypoint

__all__ = ["create_inferencer", "create_evaluator"]


def _combined(config, dataset_config):
    merged = dict(config or {})
    extra = dict(dataset_config or {})
    duplicated = merged.keys() & extra.keys()
    if duplicated:
        logging.warning(f"Duplicated keys found in two configs: `{set(duplicated)}`")
    merged.update(extra)
    return merged


def create_inferencer(net, name="topdown_heatmap", config=None, dataset_config=None, **kwargs):
    """Inference engine ``name`` (registry section "inferencer") over the evaluation network ``net``."""
    return entrypoint("inferencer", name)(net=net, config=_combined(config, dataset_config), **kwargs)


def create_evaluator(annotation_file, name="topdown", metric="AP", config=None, dataset_config=None, **kwargs):
    """Evaluation engine ``name`` (registry section "evaluator") for a COCO-format annotation file."""
    return entrypoint("evaluator", name)(annotation_file=annotation_file, metric=metric, config=_combined(config, dataset_config),
                                         **kwargs)
