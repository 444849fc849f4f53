"""sbm-bp_amd — MI355X-native belief propagation for the (degree-corrected) stochastic block model.

The product is the C-ABI library ``csrc/libsbmbp_hip.so`` (hand-written HIP kernels for gfx950,
include/sbmbp.h) and the ``bin/bp`` command line built on it. This package is the host-side
mirror of the reference's ``belief_propagation`` / ``blockmodel_t`` / ``graph_utilities``
interfaces over that C ABI (ctypes): same names, argument meaning and error behaviour, so the
parity tests read like the reference's own call sites (main.cpp:318-365).

There is no CPU fallback: importing works anywhere, but every compute entry point needs the
compiled library and a GPU and fails loudly otherwise.
"""
from sbm_bp_amd.build import build_all, lib_path  # noqa: F401
from sbm_bp_amd.capi import SbmbpError, load_library  # noqa: F401
from sbm_bp_amd.bp import (  # noqa: F401
    BeliefPropagation, Graph, blockmodel_t, bp_basic, bp_blockmodel_state, bp_conditional, bp_param_from_direct,
    bp_param_from_epsilon_c, format_infer_line, load_beliefs, load_confs, load_edge_list,
)
from sbm_bp_amd.synth import planted_partition  # noqa: F401
