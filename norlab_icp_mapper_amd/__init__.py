"""MI355X-native ICP registration core: drop-in for the hot path of norlab_icp_mapper
(``Mapper::processInput`` -> ``PM::ICPSequence``), exposed through the C ABI in include/icpmi.h.

The package holds only what the path needs: ``csrc/`` (HIP kernels + C ABI -> libicpmi.so),
``icp.py`` (host-side mirror of ``PM::ICPSequence`` over that ABI) and ``synth.py`` (the synthetic
workload of SURVEY.md 8d).  Importing the package does not load the library; constructing an
``ICPSequence`` does and fails loudly if libicpmi.so is missing.
"""
from . import synth  # noqa: F401
from .icp import (ICPSequence, ConvergenceError, InvalidField, InvalidParameter, TransformationError,  # noqa: F401
                  default_config, config_from_yaml_chain)

__all__ = ["ICPSequence", "ConvergenceError", "InvalidField", "InvalidParameter", "TransformationError",
           "default_config", "config_from_yaml_chain", "synth"]
