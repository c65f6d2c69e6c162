"""cafe-mpc_amd — MI355X-native Hybrid-Systems DDP solver behind CAFE-MPC's MultiPhaseDDP::solve() surface.

The compute path is libhsddp_hip.so (hand-written HIP for gfx950, C-ABI in include/hsddp.h).  This Python
layer only marshals descriptors; there is NO CPU fallback: if the HIP library is missing, import of the
solver fails loudly (`load_hip_library`).  The package directory name contains a hyphen, so it is loaded
through `__graft_entry__.load_package()` under the module name `cafe_mpc_amd`.
"""
import ctypes as _C
import os as _os

from . import _abi
from ._abi import Option, mhpc_ddp_setting, Solver, MODEL_WB, MODEL_SRB, MODEL_HKD, PREC_F64, PREC_F32  # noqa: F401
from . import problems  # noqa: F401
from . import launch  # noqa: F401

_HERE = _os.path.dirname(_os.path.abspath(__file__))
HIP_LIB_PATH = _os.path.join(_HERE, "libhsddp_hip.so")
# Measurement builds of the SAME HIP sources with other -D switches (make -C csrc variant NAME=x EXTRA=...; tools/ only): HSDDP_HIP_VARIANT=x selects
# variants/libhsddp_hip_x.so.  Nothing but a build of csrc/hsddp_hip.hip can be named this way.
_variant = _os.environ.get("HSDDP_HIP_VARIANT", "")
if _variant:
    if not _variant.replace("_", "").isalnum():
        raise RuntimeError(f"HSDDP_HIP_VARIANT={_variant!r}: a plain name is expected")
    HIP_LIB_PATH = _os.path.join(_HERE, "variants", f"libhsddp_hip_{_variant}.so")
_lib = None


def kernel_source_hash():
    """sha256[:16] over the kernel sources (csrc/*.hpp, *.hip): profile summaries under profiles/ record the sources they were measured on,
    bench.py only cites them while the hash still matches."""
    import hashlib
    d = _os.path.join(_HERE, "csrc"); hsh = hashlib.sha256()
    for f in sorted(_os.listdir(d)):
        if f.endswith((".hpp", ".hip")):
            hsh.update(f.encode()); hsh.update(open(_os.path.join(d, f), "rb").read())
    return hsh.hexdigest()[:16]


def load_hip_library():
    """Load and bind libhsddp_hip.so. Raises (never falls back) if it is missing or incomplete."""
    global _lib
    if _lib is None:
        if not _os.path.exists(HIP_LIB_PATH):
            raise RuntimeError(f"{HIP_LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`. "
                               "There is no CPU fallback for the product path.")
        lib = _C.CDLL(HIP_LIB_PATH)
        missing = [s for s in _abi.EXPORTS if not hasattr(lib, s)]
        if missing:
            raise RuntimeError(f"libhsddp_hip.so lacks symbols {missing}")
        _lib = _abi.bind(lib)
    return _lib


class MultiPhaseDDP(Solver):
    """Drop-in for MultiPhaseDDP<double> (HSDDPSolver/header/MultiPhaseDDP.h:22-93) on the HIP backend.

    set_multiPhaseProblem == constructor (phase descriptors), then set_initial_condition(x0[batch,n]),
    solve(option, max_cputime_ms), get_actual_cost(), get_dyn_infeasibility(), get_solver_info() ...
    Each call works on the whole batch of independent problems held by the handle.
    """

    def __init__(self, phases, batch=1, device=0, **kw):
        # precision=PREC_F32: fp32 LQ records + fp32 matrix-core sweep (kinodynamic / SRB phases), see include/hsddp.h hsddp_create_ex
        super().__init__(load_hip_library(), phases, batch=batch, device=device, **kw)
        for i, p in enumerate(phases):
            self.set_nominal(i, p["Xbar"], p["Ubar"])

    def get_actual_cost(self):
        return self.info_arrays()["actual_cost"]

    def get_dyn_infeasibility(self):
        return self.info_arrays()["dyn_feas"]

    def get_path_constraint_violation(self):
        return self.info_arrays()["max_pconstr"]

    def get_terminal_constraint_violation(self):
        return self.info_arrays()["max_tconstr"]

    def get_solver_info(self):
        a = self.info_arrays()
        return a["n_iters"], a["n_ls_iters"], a["n_reg_iters"], self.solve_time_ms()

    def get_solver_info_buffers(self, problem=0):
        """The second get_solver_info overload (MultiPhaseDDP.h:85): cost / dyn_feas / eqn_feas / ineq_feas history of one problem."""
        hst = self.get_history(problem)
        return hst["cost"], hst["dyn_feas"], hst["eqn_feas"], hst["ineq_feas"]
