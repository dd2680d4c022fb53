"""Contract of ``update_model_kwargs`` (behaviour of kgcnn/model/utils.py:69-142, stated as cases)."""
import logging

import pytest

from gcnn_keras_amd.model import utils


DEFAULTS = {"a": 1, "b": {"c": 2, "d": {"e": 3, "f": 4}}, "g": {"h": 1}, "verbose": 30}


def test_nested_merge_and_depth_limit():
    merged = utils.update_model_kwargs_logic(DEFAULTS, {"b": {"d": {"e": 9}}})
    assert merged["b"] == {"c": 2, "d": {"e": 9, "f": 4}} and merged["a"] == 1
    assert DEFAULTS["b"]["d"]["e"] == 3                                   # defaults are deep-copied, never edited
    # update_recursive = 0: dictionaries under the top level are replaced wholesale; 1: merged one level deep
    assert utils.update_model_kwargs_logic(DEFAULTS, {"b": {"d": {"e": 9}}}, 0)["b"] == {"d": {"e": 9}}
    assert utils.update_model_kwargs_logic(DEFAULTS, {"b": {"d": {"e": 9}}}, 1)["b"] == {"c": 2, "d": {"e": 9}}


def test_unknown_keys_and_type_changes(caplog):
    with pytest.raises(ValueError):
        utils.update_model_kwargs_logic(DEFAULTS, {"nope": 1})
    with caplog.at_level(logging.WARNING, logger=utils.module_logger.name):
        merged = utils.update_model_kwargs_logic(DEFAULTS, {"g": 5, "b": {"zz": 1}})
    assert merged["g"] == 5 and merged["b"]["zz"] == 1 and merged["b"]["c"] == 2
    assert "Overwriting dictionary" in caplog.text and "Unknown key" in caplog.text
    assert utils.update_model_kwargs_logic(None, None) == {}


def test_decorator_merges_sets_verbosity_and_keeps_metadata():
    @utils.update_model_kwargs(DEFAULTS)
    def make_model(**kwargs):
        """doc"""
        return kwargs

    out = make_model(b={"c": 7}, verbose=10)
    assert out["b"] == {"c": 7, "d": {"e": 3, "f": 4}} and out["a"] == 1
    assert utils.module_logger.level == 10                                # taken from the merged `verbose`
    assert make_model.__name__ == "make_model" and make_model.__doc__ == "doc"
    with pytest.raises(ValueError):
        make_model(other=1)
    utils.module_logger.setLevel(logging.WARNING)
