"""MI355X-native hot path of TrafficBots (scene encoders + closed-loop multi-agent rollout) behind the reference's task-module
surface.  `trafficbots_amd.WaymoMotion` mirrors `pl_modules.waymo_motion.WaymoMotion`; `trafficbots_amd.instantiate(cfg)` is the
`hydra.utils.instantiate` call of `src/run.py:34-36` for a config dict shaped like `configs/model/traffic_bots.yaml`."""


def __getattr__(name):  # lazy: importing the package alone must not pull torch / the HIP library in
    if name == "WaymoMotion":
        from .waymo_motion import WaymoMotion

        return WaymoMotion
    if name == "instantiate":
        from .config import instantiate

        return instantiate
    raise AttributeError(name)
