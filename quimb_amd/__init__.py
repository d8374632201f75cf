"""quimb_amd -- an MI355X (gfx950) tensor-network contraction backend that
plugs in at quimb's autoray array-backend boundary.

The top-level module *is* the autoray backend namespace: ``autoray.do(name,
x, like="quimb_amd")`` resolves ``quimb_amd.<name>``, and
``autoray.infer_backend`` maps ``quimb_amd.Array`` instances to ``"quimb_amd"``
because the class is exported from here (see ``autoray_backend.register``).

Importing this package never touches the GPU; the HIP library and device are
bound on first use and there is no CPU fallback.
"""

from .array import Array, asarray, to_numpy
from .ops import (  # noqa: F401  (autoray resolves these by name)
    abs, absolute, absmax, add, amax, amin, array, astype, concatenate, conj, conjugate, diagonal, divide, dot,
    einsum, einsum_pair, exp, expand_dims, eye, fuse, imag, log, log10, matmul, max, min, multiply, ndim, negative,
    norm_fro, ones, ravel, real, reshape, shape, size, sqrt, squeeze, subtract, sum, take, tensordot, trace,
    transpose, true_divide, zeros,
    kron, mean, moveaxis, outer, power, square, stack, swapaxes, vdot, implementation_pair,
)
from . import linalg  # noqa: F401  (do("linalg.svd" / "linalg.qr" / "linalg.eigh" / "linalg.norm"))
from .contract import (  # noqa: F401
    ContractExpression, Tensor, array_contract, array_contract_expression, array_contract_path,
    array_contract_tree, contract_backend, contract_strategy, get_contract_backend,
    get_contract_strategy, get_tensor_linop_backend, set_contract_backend, set_contract_strategy,
    set_tensor_linop_backend, tensor_contract, tensor_linop_backend,
)
from .executor import TreeExecutor
from .options import Options, get_options, set_options
from .options import options as exec_options
from .linop import TNLinearOperator
from .eigsolve import eigh_lanczos
from .microtree import MicroTree
from .boundary import contract_boundary_2d
from .dmrg import DMRG2, mpo_ham_heis
from .split import array_split, tensor_split
from .circuit import Circuit, CircuitMPS
from .network import TensorNetwork
from .pathfind import (bisection_ssa, find_path, find_slices, fused_pair_count, geometry_hash, greedy_path, modeled_time,
                       quadrant_path_2d, random_greedy, set_tree_cache, sweep_path_2d)
from .tree import ContractionTree
from .twosided import TwoSidedContraction
from . import quadrants
from .quadrants import QuadrantRank, QuadrantSharding, contract_quadrants
from . import rangeslice
from .rangeslice import RangeSliced, RangeSlicedExecutor, contract_range_sliced, find_range_slices
from .device import HipDevice, default_device

# the class must report the top-level module for autoray's backend inference
Array.__module__ = "quimb_amd"

__version__ = "0.1.0"
