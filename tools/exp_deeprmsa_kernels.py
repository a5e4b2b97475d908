import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from conftest import DEEPRMSA_NODE_PROBS, load_topology
from optical_rl_gym_amd import BatchedDeepRMSAEnv
topo = load_topology("nsfnet_chen_5-paths_6-modulations")
for kern in ("wave", "group", "auto"):
    env = BatchedDeepRMSAEnv(topo, 32768, num_spectrum_resources=320, j=1, mean_service_holding_time=7.5, mean_service_inter_arrival_time=1/12.0,
                             node_request_probabilities=DEEPRMSA_NODE_PROBS, episode_length=50, seed=10, step_kernel=kern)
    obs = torch.empty((32768, env.obs_dim), dtype=torch.float64, device="cuda")
    def step():
        env.run("deeprmsa_sap_ff", 1, auto_reset=True); env.observation(out=obs)
    for _ in range(300): step()
    env.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000): step()
    env.synchronize(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(kern, "ms/step %.4f" % (dt / 2000 * 1e3), env.last_kernel(), "running", env.num_running().mean())
    env.close()
