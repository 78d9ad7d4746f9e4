# dev tool: where do GPU and oracle poses differ?
import sys, os
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
sys.path.insert(0, os.path.join(R_, "tests"))
import numpy as np
import oracle_lib as O
if os.path.exists(os.path.join(R_, "oracle/_build/liboracle_dbg.so")):
    O._SO = os.path.join(R_, "oracle/_build/liboracle_dbg.so")
from rmcv_amd import Context, default_pnp_config
from test_oracle_pnp import project, rodrigues
O.set_math_mode(0)
ctx = Context(device=0, max_frames=1)
ctx.pnp_load()
ocfg = O.default_pnp_config()
rng = np.random.default_rng(5)
arm = np.zeros(1, O.ARMOUR)
R = rodrigues(rng.uniform(-0.9, 0.9, 3))
t = np.array([rng.uniform(-400, 400), rng.uniform(-300, 300), rng.uniform(500, 6000)])
arm[0]["vertices"] = project(R, t, ocfg)
got = ctx.locate_armours(arm)
sys.stdout.flush()
want = O.locate_armours(arm, ocfg)
print(got[1], want[1])
