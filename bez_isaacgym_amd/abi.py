"""ctypes mirror of include/bez_sim.h (the C ABI of libbez_sim.so).

Only declarations live here: struct layout, enums, and the default `BezSimConfig` for
`bez_kick` (values: bez_isaacgym/cfg/task/bez_kick.yaml:11-147, kick_env.py:322-329 of the
reference).  Loading / calling the library is in `bez_isaacgym_amd.sim`.
"""
import ctypes as C
import math

ABI_VERSION = 5
NUM_OBS = 54
NUM_ACTIONS = 18
NUM_DOFS = 18
NUM_BODIES = 22
NUM_ACTORS = 2
NUM_LINKS = 19

FLAG_IMU_PREV_ALIAS = 1
FLAG_CF_WITH_FRICTION = 2
FLAG_CF_LAST_SUBSTEP = 4
FLAG_NO_SELF_COLLISION = 8
FLAG_CLEATS = 16
FLAG_BOX_ASSET = 32
FLAG_HARD_CONTACT = 64
FLAG_LEAN_STEP = 128
FLAG_OBS_NOISE_IN_STEP = 256
FLAG_TGS_SOLVER = 512
FLAG_ANKLE_STOP = 1024          # scenario harness: calf <-> foot-plate contact (one-env-per-lane kernel, stl asset without cleats)
FLAG_ALL_GROUND_SHAPES = 2048   # scenario harness: ground contact at every collision shape's corners (same)
FLAG_FIX_BASE = 4096            # urdfAsset.fixBaseLink: the torso welded to the world
TASK_KICK, TASK_WALK, TASK_ORIENT = 0, 1, 2
TASK_IDS = {"bez_kick": TASK_KICK, "bez_walk": TASK_WALK, "bez_orient": TASK_ORIENT}

(TENSOR_ROOT_STATE, TENSOR_DOF_STATE, TENSOR_RIGID_BODY_STATE, TENSOR_NET_CONTACT_FORCE, TENSOR_OBS,
 TENSOR_REW, TENSOR_RESET, TENSOR_PROGRESS, TENSOR_TIMEOUT, TENSOR_DOF_TARGET, TENSOR_PREV_LIN_VEL,
 TENSOR_FEET, TENSOR_GOAL, TENSOR_RANDOMIZE_BUF, TENSOR_DR_NOISE, TENSOR_COUNT) = range(16)
DTYPE_F32, DTYPE_I64 = 0, 1
(PARAM_FRICTION, PARAM_KP_SCALE, PARAM_KD_SCALE, PARAM_MASS_SCALE, PARAM_GRAVITY, PARAM_DOF_LOWER, PARAM_DOF_UPPER,
 PARAM_COUNT) = range(8)
PARAM_WIDTH = {PARAM_FRICTION: 1, PARAM_KP_SCALE: 18, PARAM_KD_SCALE: 18, PARAM_MASS_SCALE: 19, PARAM_GRAVITY: 3,
               PARAM_DOF_LOWER: 18, PARAM_DOF_UPPER: 18}


class BezSimConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("num_envs", C.c_int32),
        ("substeps", C.c_int32),
        ("max_episode_length", C.c_int32),
        ("dt", C.c_float),
        ("gravity", C.c_float * 3),
        ("kp", C.c_float),
        ("kd", C.c_float),
        ("armature", C.c_float),
        ("effort", C.c_float),
        ("vel_limit", C.c_float),
        ("joint_friction", C.c_float),
        ("plane_friction", C.c_float),
        ("clip_actions", C.c_float),
        ("bez_init", C.c_float * 7),
        ("ball_init", C.c_float * 7),
        ("goal", C.c_float * 2),
        ("contact_kn", C.c_float),
        ("contact_cn", C.c_float),
        ("contact_ct", C.c_float),
        ("contact_veps", C.c_float),
        ("limit_k", C.c_float),
        ("limit_d", C.c_float),
        ("jfric_veps", C.c_float),
        ("ball_ang_damping", C.c_float),
        ("self_kn", C.c_float),
        ("self_cn", C.c_float),
        ("ball_kn", C.c_float),
        ("ball_cn", C.c_float),
        ("tune", C.c_float * 24),
        ("task", C.c_int32),
        ("goal_angle", C.c_float),
        ("flags", C.c_uint32),
        ("seed", C.c_uint64),
        ("env_id_offset", C.c_int64),
    ]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out


class BezDrRange(C.Structure):
    _fields_ = [("a", C.c_float), ("b", C.c_float), ("enabled", C.c_int32), ("schedule_steps", C.c_int32)]


class BezDrConfig(C.Structure):
    """device-side domain randomisation (include/bez_sim.h); one BezDrRange per entry of bez_kick.yaml:151-219"""
    _fields_ = [("frequency", C.c_int32), ("friction_buckets", C.c_int32), ("friction", BezDrRange), ("stiffness", BezDrRange),
                ("damping", BezDrRange), ("lower", BezDrRange), ("upper", BezDrRange), ("gravity", BezDrRange),
                ("observations", BezDrRange), ("actions", BezDrRange)]


def dr_config_from_params(dr_params):
    """randomization_params of cfg/task/bez_kick.yaml:151-219 -> BezDrConfig.  Distribution / operation per parameter are the ones
    that file uses; anything else is refused rather than silently reinterpreted."""
    d = BezDrConfig()
    d.frequency = int(dr_params.get("frequency", 1))

    def put(dst, attr, dist, op):
        if attr is None:
            return
        if attr.get("distribution") != dist or attr.get("operation") != op:
            raise ValueError("domain randomisation: %s must be %s / %s as in bez_kick.yaml (got %s / %s)" % (dst, dist, op, attr.get("distribution"), attr.get("operation")))
        r = getattr(d, dst)
        r.a, r.b = float(attr["range"][0]), float(attr["range"][1])
        r.enabled = 1
        sch = attr.get("schedule")
        if sch not in (None, "linear"):
            raise ValueError("domain randomisation: only `schedule: linear` is supported (%s)" % dst)
        r.schedule_steps = int(attr["schedule_steps"]) if sch == "linear" else 0
    put("observations", dr_params.get("observations"), "gaussian", "additive")
    put("actions", dr_params.get("actions"), "gaussian", "additive")
    put("gravity", (dr_params.get("sim_params") or {}).get("gravity"), "gaussian", "additive")
    ap = ((dr_params.get("actor_params") or {}).get("bez") or {})
    fr = (ap.get("rigid_shape_properties") or {}).get("friction")
    put("friction", fr, "uniform", "scaling")
    d.friction_buckets = int((fr or {}).get("num_buckets", 0) or 0)
    dp = ap.get("dof_properties") or {}
    put("stiffness", dp.get("stiffness"), "uniform", "scaling")
    put("damping", dp.get("damping"), "uniform", "scaling")
    put("lower", dp.get("lower"), "gaussian", "additive")
    put("upper", dp.get("upper"), "gaussian", "additive")
    return d


# Contact / limit model constants of this build (no reference counterpart; DESIGN.md "Physics model")
CONTACT_DEFAULTS = dict(contact_kn=2.0e4, contact_cn=20.0, contact_ct=1.0e3, contact_veps=0.01,
                        limit_k=200.0, limit_d=2.0, jfric_veps=0.1, ball_ang_damping=0.5,
                        self_kn=2.0e4, self_cn=5.0, ball_kn=0.0, ball_cn=0.0)


def default_config(num_envs=4096, seed=42, env_id_offset=0):
    """bez_kick defaults; must agree field-for-field with bez_sim_default_config() in the library."""
    c = BezSimConfig()
    c.abi_version = ABI_VERSION
    c.num_envs = int(num_envs)
    c.substeps = 2
    c.dt = 0.01667
    c.max_episode_length = int(15.0 / 0.01667 + 0.5)  # kick_env.py:127 -> 900
    c.gravity[:] = [0.0, 0.0, -9.81]
    c.kp, c.kd = 100.0, 7.5
    c.armature = 0.001
    c.effort = 2.5
    c.vel_limit = 2.0 * math.pi
    c.joint_friction = 0.1
    c.plane_friction = 1.0
    c.clip_actions = 3.9
    c.bez_init[:] = [0.0, 0.0, 0.34, 0.0, 0.0, 0.0, 1.0]
    c.ball_init[:] = [0.175, 0.0, 0.1, 0.0, 0.0, 0.0, 1.0]
    c.goal[:] = [1.5, 0.0]
    for k, v in CONTACT_DEFAULTS.items():
        setattr(c, k, v)
    c.flags = FLAG_IMU_PREV_ALIAS
    c.seed = int(seed)
    c.env_id_offset = int(env_id_offset)
    return c


def refuse_unmodelled(cfg):
    """Config keys the reference forwards to Isaac Gym that this build's rigid-body step does not model: a non-default value raises
    instead of being read and ignored (the defaults of bez_kick.yaml are what the kernels implement; fixBaseLink IS modelled: BEZ_FLAG_FIX_BASE).
    kick_env.py:250-256 (plane), :283-294 (asset options)."""
    env = cfg["env"]
    ua, plane = env.get("urdfAsset", {}), env.get("plane", {})

    def no(cond, key, value, why):
        if cond:
            raise ValueError("task config: %s = %r is not modelled by the HIP simulator (%s); only the value of the reference's "
                             "yaml is implemented" % (key, value, why))
    no(bool(ua.get("disable_gravity", False)), "env.urdfAsset.disable_gravity", ua.get("disable_gravity"),
       "gravity acts on every body; set sim.gravity to zero instead")
    for k in ("angular_damping", "linear_damping"):
        no(float(ua.get(k, 0.0)) != 0.0, "env.urdfAsset." + k, ua.get(k), "per-body velocity damping is not part of the step")
    no(float(plane.get("restitution", 0.0)) != 0.0, "env.plane.restitution", plane.get("restitution"), "contacts are inelastic")
    if "staticFriction" in plane and "dynamicFriction" in plane:
        no(float(plane["staticFriction"]) != float(plane["dynamicFriction"]), "env.plane.staticFriction", plane["staticFriction"],
           "one Coulomb coefficient (dynamicFriction) serves for both")
    # vec_task.py:90 reads env.controlFrequencyInv (the yaml's env.control.controlFrequencyInv is never read by the reference):
    # k simulate calls per env step, taken by KickEnv._fused_step through the split entry points
    cfi = int(env.get("controlFrequencyInv", 1))
    no(cfi < 1, "env.controlFrequencyInv", cfi, "must be >= 1")


def config_from_task_cfg(cfg, seed=42, env_id_offset=0, strict_reference_quirks=True, task="bez_kick"):
    """Build a BezSimConfig from the task config dict (the structure of cfg/task/bez_{kick,walk,orient}.yaml)."""
    env, sim = cfg["env"], cfg["sim"]
    refuse_unmodelled(cfg)
    c = default_config(int(env["numEnvs"]), seed=seed, env_id_offset=env_id_offset)
    c.task = TASK_IDS[task]
    c.goal_angle = float(env["goalState"].get("goal_angle", 0.0))
    c.substeps = int(sim.get("substeps", 2))
    c.dt = float(sim["dt"])
    c.max_episode_length = int(float(env["learn"]["episodeLength_s"]) / float(sim["dt"]) + 0.5)
    c.gravity[:] = [float(x) for x in sim["gravity"]]
    c.kp = float(env["control"]["stiffness"])
    c.kd = float(env["control"]["damping"])
    c.armature = float(env["urdfAsset"]["armature"])
    c.plane_friction = float(env["plane"]["dynamicFriction"])
    c.clip_actions = float(env.get("clipActions", float("inf")))
    c.bez_init[:] = [float(x) for x in env["bezInitState"]["pos"] + env["bezInitState"]["rot"]]
    if "ballInitState" in env:  # bez_walk / bez_orient have no ball actor
        c.ball_init[:] = [float(x) for x in env["ballInitState"]["pos"] + env["ballInitState"]["rot"]]
    c.goal[:] = [float(x) for x in env["goalState"]["goal"]]
    for k in CONTACT_DEFAULTS:
        if k in sim.get("bez", {}):
            setattr(c, k, float(sim["bez"][k]))
    for k in ("effort", "vel_limit", "joint_friction"):   # kick_env.py:322-329 hard-codes these; sim.bez.<k> overrides them for experiments
        if k in sim.get("bez", {}):
            setattr(c, k, float(sim["bez"][k]))
    c.flags = FLAG_IMU_PREV_ALIAS if strict_reference_quirks else 0
    if env.get("asset", {}).get("cleats", False):
        c.flags |= FLAG_CLEATS
    if not env.get("asset", {}).get("stl", True):  # kick_env.py:266-276: soccerbot_box*.urdf
        c.flags |= FLAG_BOX_ASSET
    if env.get("urdfAsset", {}).get("fixBaseLink", False):  # kick_env.py:287
        c.flags |= FLAG_FIX_BASE
    return c
