import sys; sys.path[:0]=["/root/repo","/root/repo/tests","/root/repo/oracle"]
from conftest import load_topology
from optical_rl_gym_amd import BatchedRMSAEnv
for st in ("full","counters"):
    e=BatchedRMSAEnv(load_topology("nsfnet_chen_5-paths_6-modulations"),64,num_spectrum_resources=320,load=50,mean_service_holding_time=25,stats_level=st)
    print(st, e.launch_info()); e.close()
