"""Environment ids of the reference's registry (``optical_rl_gym/__init__.py:3-31``) for the environments on the hot
path: ``RMSA-v0``, ``DeepRMSA-v0``, ``PhyRMSA-v0``.  :func:`make` resolves an id to the single-env view class and
constructs it (``gym.make(id, **env_args)`` in the reference's scripts, e.g. ``tests/test_rmsa_threads_us.py:64``);
:func:`register_with_gym` adds the same ids to an installed ``gymnasium`` / ``gym`` registry under this package's entry
points (neither is required: the views do not import gym)."""
import importlib

ENV_IDS = {
    "RMSA-v0": ("optical_rl_gym_amd.envs", "RMSAEnv"),            # optical_rl_gym/__init__.py:8-11
    "DeepRMSA-v0": ("optical_rl_gym_amd.envs", "DeepRMSAEnv"),    # :13-16
    "PhyRMSA-v0": ("optical_rl_gym_amd.phy_env", "PhyRMSAEnv"),   # :28-31
}


def env_class(env_id: str):
    try:
        module, name = ENV_IDS[env_id]
    except KeyError:
        raise KeyError(f"unknown environment id {env_id!r}; known: {sorted(ENV_IDS)}") from None
    return getattr(importlib.import_module(module), name)


def make(env_id: str, **kwargs):
    """``gym.make(env_id, **kwargs)`` for the three ids above."""
    return env_class(env_id)(**kwargs)


def register_with_gym():
    """Register the ids with gymnasium or gym when one of them is installed; returns the module used or None."""
    for modname in ("gymnasium", "gym"):
        try:
            mod = importlib.import_module(modname)
        except ImportError:
            continue
        from_registry = getattr(mod.envs, "registry", {})
        for env_id, (module, name) in ENV_IDS.items():
            if env_id not in from_registry:
                mod.envs.registration.register(id=env_id, entry_point=f"{module}:{name}")
        return mod
    return None
