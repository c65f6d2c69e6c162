"""Fixture generator (run in the build container, needs /root/reference): copies DATA the reference ships — the first N
samples of two gait reference files and the three MHPC settings files — into tests/golden/cafe_tree/ with the
directory layout of a CAFE-MPC checkout, so that cafe_mpc_amd.builder.build_from_tree can be exercised where the
reference tree does not exist (the GPU box).  No reference source code is copied."""
import os
import shutil

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cafe_tree")
N = 130   # samples kept: plan horizon 0.75 s at dt = 0.01 needs 77, plus 0.5 s of receding-horizon shifts


def trim(src, dst, n):
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    out, cnt = [], 0
    for line in open(src):
        out.append(line)
        if "status_dur" in line:
            cnt += 1
        elif cnt and out[-2].startswith("status_dur") and cnt >= n:
            break
    open(dst, "w").writelines(out)


for gait in ("bound", "trot/dynfeas"):
    trim(os.path.join(REF, "Reference/Data", gait, "quad_reference.csv"), os.path.join(OUT, "Reference/Data", gait, "quad_reference.csv"), N)
for f in ("mhpc_config.info", "cost_weights_regular.JSON", "constraint_params_regular.info", "ddp_setting.info"):
    os.makedirs(os.path.join(OUT, "MHPC/settings"), exist_ok=True)
    shutil.copy(os.path.join(REF, "MHPC/settings", f), os.path.join(OUT, "MHPC/settings", f))
    os.chmod(os.path.join(OUT, "MHPC/settings", f), 0o644)
for f in ("constraint_params.info", "ddp_setting.info"):
    os.makedirs(os.path.join(OUT, "HKDMPC/settings"), exist_ok=True)
    shutil.copy(os.path.join(REF, "HKDMPC/settings", f), os.path.join(OUT, "HKDMPC/settings", f))
    os.chmod(os.path.join(OUT, "HKDMPC/settings", f), 0o644)
print("wrote", OUT)
