/* bez_sim.h -- C ABI of the MI355X-native Bez `bez_kick` simulator (libbez_sim.so).
 *
 * This is the drop-in boundary for the hot path VecTask.step():
 *   reference  bez_isaacgym/tasks/base/vec_task.py:303-349   (step orchestration)
 *              bez_isaacgym/tasks/kick_env.py:410-438,749-850 (pre/post physics, obs, reset)
 * The reference has no FFI of its own: everything below the Python `VecTask` goes into the
 * closed Isaac Gym binary through the `gymapi` tensor API.  Each entry point therefore cites
 * the gym call (and its call site in the reference) that it replaces.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `stream` is a hipStream_t passed as void*
 *     (NULL = the null stream).  All work is enqueued on that stream; nothing synchronises
 *     unless stated.
 *   - every pointer named *_dev is device memory on the sim's GPU, borrowed for the call.
 *   - every function returns 0 on success, <0 on error (message: bez_sim_last_error()).
 *   - the sim owns all state buffers for its lifetime; bez_sim_get_tensor() exposes them
 *     zero-copy (gymtorch.wrap_tensor equivalent, kick_env.py:155-157).
 *   - Isaac-layout tensors (row-major AoS, fp32, per env: A actors, B bodies, 18 DOFs):
 *       ROOT_STATE        (N*A, 13)  pos3 quat_xyzw4 linvel3 angvel3     kick_env.py:143
 *       DOF_STATE         (N*18, 2)  pos vel                             kick_env.py:144
 *       RIGID_BODY_STATE  (N*B, 13)                                      kick_env.py:145
 *       NET_CONTACT_FORCE (N*B, 3)                                       kick_env.py:146
 *     bez_kick: A = 2 (robot, ball), B = 22 (21 robot bodies + ball); with the cleats asset (BEZ_FLAG_CLEATS) B = 30;
 *     bez_walk / bez_orient have no ball actor: A = 1, B = 21 (29 with cleats); OBS is (N,54) for bez_kick, (N,52) otherwise.
 *     They are *materialised on demand* by bez_sim_refresh_tensor (gym.refresh_*_tensor);
 *     the simulator's own state is SoA ([field][env], one env per lane).
 */
#ifndef BEZ_SIM_H
#define BEZ_SIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BEZ_SIM_ABI_VERSION 5

#define BEZ_NUM_OBS 54       /* bez_kick; bez_walk / bez_orient: 52 (no ball tail)  walk_env.py:104 */
#define BEZ_NUM_OBS_WALK 52
#define BEZ_NUM_ACTIONS 18
#define BEZ_NUM_DOFS 18
#define BEZ_NUM_BODIES 22 /* 21 robot + ball */
#define BEZ_NUM_ACTORS 2

/* flags */
#define BEZ_FLAG_IMU_PREV_ALIAS 1u /* quirk Q1 (kick_env.py:930,441): prev_lin_vel aliases the live \
                                      velocity, so the finite difference is identically 0 */

#define BEZ_FLAG_CF_LAST_SUBSTEP 4u   /* NET_CONTACT_FORCE = last substep only; default: mean over the substeps of the \
                                        control step (physx.contact_collection 2 = CC_ALL_SUBSTEPS, bez_kick.yaml:147) */
#define BEZ_FLAG_NO_SELF_COLLISION 8u /* disable leg<->leg contact (the reference enables self-collision) */
#define BEZ_FLAG_CF_WITH_FRICTION 2u /* NET_CONTACT_FORCE rows include friction; default off: Isaac Gym reports the \
                                        normal contact impulses only [ext] (checkpoint obs statistics) */

#define BEZ_FLAG_CLEATS 16u /* asset.cleats: True -> soccerbot_stl_sensor.urdf: 29 robot bodies, per-cleat contact rows \
                               13:17 / 25:29 and compute_feet_sensors_cleats (kick_env.py:187-191,267-276,1044-1069) */
#define BEZ_FLAG_BOX_ASSET 32u /* asset.stl: False -> soccerbot_box.urdf / soccerbot_box_sensor.urdf (kick_env.py:266-276): the stl \
                                  asset's dynamics with the URDF's own torso / head / forearm collision boxes (upper-body ground \
                                  points, ball <-> torso box); with BEZ_FLAG_CLEATS also that URDF's right ankle joint origin */

#define BEZ_FLAG_HARD_CONTACT 64u /* rigid contact: velocity-level impulses with Coulomb stiction and restitution 0 (projected Gauss-Seidel on \
                                     the articulated-body impulse responses) instead of the implicit spring-dampers; knobs in `tune`. \
                                     ORACLE ONLY (an experiment that did not earn a kernel): bez_sim_create / \
                                     bez_sim_set_flags of libbez_sim.so refuse it with rc -5 */
#define BEZ_FLAG_TGS_SOLVER 512u /* TGS-shaped unified substep: drives, joint friction, joint limits, joint speed limit and every contact \
                                    as clamped impulse rows relaxed together over posIters sub-steps of dt / posIters (+ velIters), the \
                                    solver shape PhysX is configured with (bez_kick.yaml:128-147); knobs in `tune[8..23]`.  ORACLE ONLY \
                                    (DESIGN.md 3.2 / 6.1): libbez_sim.so refuses it with rc -5 */

#define BEZ_FLAG_ANKLE_STOP 1024u /* same-leg calf <-> foot-plate contact.  `create_actor(..., collision_filter 0)` (kick_env.py:365-366) makes \
                                     PhysX collide every non-adjacent pair of the robot's shapes, and one such pair limits the robot's motion \
                                     all the time: the bottom corners of the calf box (soccerbot_stl.urdf:232-236) meet the top face of the \
                                     3 mm foot plate (:272-276) once the flexed ankle also rolls by 0.3-0.45 rad -- well inside the roll \
                                     joint's own +-0.785 rad.  The gap is a function of the two ankle angles alone, so the contact is a \
                                     coupled limit of those two joints: an implicit spring-damper along the gap's gradient.  A scenario-harness \
                                     variant (the reference policy's sim-to-sim does not move with it): stepped by the \
                                     one-env-per-lane kernel, stl asset without cleats only (any other asset: rc -5) */

#define BEZ_FLAG_ALL_GROUND_SHAPES 2048u /* ground contact at the corners of EVERY collision shape of soccerbot_stl.urdf (hip, thigh, calf, ankle, \
                                            forearm, neck, head boxes / mesh bounds), not only the foot corners and the upper-body guard points. \
                                            For the scenario harness (the get-up tables put knees, hips and forearms on the ground, \
                                            soccer_trajectories.py:56-91; nothing on the bez_kick path touches the ground with them before its \
                                            fall reset): stepped by the one-env-per-lane kernel, stl asset without cleats only (else rc -5) */

#define BEZ_FLAG_FIX_BASE 4096u /* urdfAsset.fixBaseLink: True (kick_env.py:287, bez_kick.yaml:83): the torso is welded to the world -- its spatial \
                                   acceleration is zero instead of the 6 x 6 root solve's, its twist stays zero, the pose stays where reset put it; \
                                   joints, contacts and the ball are unchanged.  (The reference's per-DOF sweep, test/test_kick_env.py:142-186, \
                                   notes "better when fixBaseLink = True".)  A test configuration: stepped by the one-env-per-lane kernel \
                                   (3.3 x 10^7 env-steps/s at 4096 envs), not by the 8-role-wave kernel of the default path. */

#define BEZ_FLAG_LEAN_STEP 128u /* bez_sim_step / bez_sim_step_many keep only what the rollout reads (state, obs, reward, reset / progress / \
                                   timeout, DOF targets): the stores of the NET_CONTACT_FORCE rows, FEET and PREV_LIN_VEL -- 308 B of the \
                                   912 B an env-step writes -- are skipped, and those three tensors then hold the values of the last call \
                                   without the flag.  The feet flags inside the observation are unaffected (computed from the in-kernel \
                                   forces).  Needs BEZ_FLAG_IMU_PREV_ALIAS (prev_lin_vel is then never read back). */

#define BEZ_FLAG_OBS_NOISE_IN_STEP 256u /* with a randomisation that has observation noise (bez_sim_set_randomization): the post-physics \
                                          part of bez_sim_step / bez_sim_post_physics writes the NOISY observations itself (the noise \
                                          rides on the kernel's copy-out, bit-identical to what bez_sim_add_dr_noise(OBS, OBS, 0) adds), \
                                          and that call on the observation tensor becomes a no-op: one launch less per control step */

/* Tasks sharing the robot, the physics and the tensor API; they differ in the post-physics logic (observation tail,
 * reward, reset conditions, goal sampling) and in the ball actor (bez_kick only). */
#define BEZ_TASK_KICK 0   /* tasks/kick_env.py    54 obs, ball + goal point                              */
#define BEZ_TASK_WALK 1   /* tasks/walk_env.py    52 obs, no ball, goal xy ~ U(-2,2)^2 redrawn at reset  */
#define BEZ_TASK_ORIENT 2 /* tasks/orient_env.py  52 obs, no ball, goal heading                          */

typedef struct BezSimConfig {
  int32_t abi_version; /* must be BEZ_SIM_ABI_VERSION */
  int32_t num_envs;    /* cfg["env"]["numEnvs"]                         vec_task.py:84 */
  int32_t substeps;    /* cfg["sim"]["substeps"]                        vec_task.py:430 */
  int32_t max_episode_length; /* int(episodeLength_s/dt+0.5)            kick_env.py:127 */
  float dt;            /* cfg["sim"]["dt"]                              vec_task.py:427 */
  float gravity[3];    /* cfg["sim"]["gravity"]                         vec_task.py:439 */
  float kp, kd;        /* control.stiffness / damping                   kick_env.py:324-325 */
  float armature;      /* urdfAsset.armature                            kick_env.py:326 */
  float effort;        /* 2.5 N*m all DOFs                              kick_env.py:329 */
  float vel_limit;     /* 2*pi rad/s                                    kick_env.py:327 */
  float joint_friction;/* 0.1                                           kick_env.py:328 */
  float plane_friction;/* plane.dynamicFriction                         kick_env.py:254 */
  float clip_actions;  /* env.clipActions                               vec_task.py:98,317 */
  float bez_init[7];   /* bezInitState pos + rot(xyzw)                  kick_env.py:60-64 */
  float ball_init[7];  /* ballInitState pos + rot                       kick_env.py:67-71 */
  float goal[2];       /* goalState.goal                                kick_env.py:74,159 */
  /* Contact / limit model of this build (PhysX's TGS contact solve has no source here; see
   * DESIGN.md "Physics model").  All forces are spring-dampers integrated implicitly. */
  float contact_kn;    /* normal stiffness per contact point  [N/m]   */
  float contact_cn;    /* normal damping                      [N*s/m] */
  float contact_ct;    /* max tangential (stick) viscosity    [N*s/m] */
  float contact_veps;  /* Coulomb regularisation speed        [m/s]   */
  float limit_k;       /* joint-limit spring                  [N*m/rad]   */
  float limit_d;       /* joint-limit damper                  [N*m*s/rad] */
  float jfric_veps;    /* joint-friction regularisation speed [rad/s] */
  float ball_ang_damping; /* Isaac asset default angular_damping 0.5 [ext] for the ball actor */
  float self_kn;       /* leg<->leg self-collision (kick_env.py:365-366, filter 0): spring [N/m]   */
  float self_cn;       /*                                                     damper [N*s/m] */
  float ball_kn;       /* ball <-> ground / ball <-> robot contact spring [N/m]; 0 = contact_kn */
  float ball_cn;       /* and damper [N*s/m]; 0 = contact_cn.  PhysX restitution is 0 (bez_kick.yaml:16): critical damping of the 0.3 kg
                          ball against contact_kn is 2 sqrt(kn m) = 155 N*s/m */
  float tune[24];      /* knobs of the oracle-only solver variants (BEZ_FLAG_HARD_CONTACT: [0..7], BEZ_FLAG_TGS_SOLVER: [8..23];
                          oracle/bez_oracle.c documents the slots); 0 = default.  libbez_sim.so refuses a non-zero entry (rc -5) */
  int32_t task;        /* BEZ_TASK_*: which env logic POST runs (tasks/__init__.py:10-16)                */
  float goal_angle;    /* bez_orient: env.goalState.goal_angle                       orient_env.py:61 */
  uint32_t flags;      /* BEZ_FLAG_* */
  uint64_t seed;       /* reset-noise stream key (config.yaml:11 seed: 42) */
  int64_t env_id_offset; /* global id of local env 0; reset noise is keyed by GLOBAL env id so
                            results do not depend on how envs are sharded over GPUs */
} BezSimConfig;

/* Fills `cfg` with the bez_kick.yaml / kick_env.py defaults for `num_envs` environments. */
int bez_sim_default_config(BezSimConfig* cfg, int32_t num_envs);

typedef struct BezSim BezSim;

enum BezTensor {
  BEZ_TENSOR_ROOT_STATE = 0,        /* f32 (N*2,13)   acquire_actor_root_state_tensor   kick_env.py:143 */
  BEZ_TENSOR_DOF_STATE = 1,         /* f32 (N*18,2)   acquire_dof_state_tensor          kick_env.py:144 */
  BEZ_TENSOR_RIGID_BODY_STATE = 2,  /* f32 (N*22,13)  acquire_rigid_body_state_tensor   kick_env.py:145 */
  BEZ_TENSOR_NET_CONTACT_FORCE = 3, /* f32 (N*22,3)   acquire_net_contact_force_tensor  kick_env.py:146 */
  BEZ_TENSOR_OBS = 4,               /* f32 (N,54)     obs_buf       vec_task.py:235 */
  BEZ_TENSOR_REW = 5,               /* f32 (N)        rew_buf       vec_task.py:239 */
  BEZ_TENSOR_RESET = 6,             /* i64 (N)        reset_buf     vec_task.py:241 (init 1) */
  BEZ_TENSOR_PROGRESS = 7,          /* i64 (N)        progress_buf  vec_task.py:245 */
  BEZ_TENSOR_TIMEOUT = 8,           /* i64 (N)        timeout_buf   vec_task.py:243 */
  BEZ_TENSOR_DOF_TARGET = 9,        /* f32 (N,18)     PD position targets */
  BEZ_TENSOR_PREV_LIN_VEL = 10,     /* f32 (N,3)      prev_lin_vel  kick_env.py:183 */
  BEZ_TENSOR_FEET = 11,             /* f32 (N,8)      self.feet     kick_env.py:185 */
  BEZ_TENSOR_GOAL = 12,             /* f32 (N,2)      self.goal     walk_env.py:143 (bez_walk / bez_orient: redrawn at reset) */
  BEZ_TENSOR_RANDOMIZE_BUF = 13,    /* i64 (N)        randomize_buf vec_task.py:247 (device-side domain randomisation) */
  BEZ_TENSOR_DR_NOISE = 14,         /* f32 (4)        mean / std of the observation noise, mean / std of the action noise as the schedule
                                                      currently has them (vec_task.py:544-618): the caller's noise lambdas read them on the device */
  BEZ_TENSOR_COUNT = 15
};
enum BezDtype { BEZ_DTYPE_F32 = 0, BEZ_DTYPE_I64 = 1 };

/* gym.create_sim + create_env/create_actor loop + prepare_sim + allocate_buffers
 * (vec_task.py:174-193, kick_env.py:240-408) followed by KickEnv.__init__'s reset_idx(all)
 * (kick_env.py:238): every env starts from its first reset draw with reset_buf = 0. */
int bez_sim_create(const BezSimConfig* cfg, int device_id, BezSim** out);
int bez_sim_destroy(BezSim* sim);
const char* bez_sim_last_error(const BezSim* sim); /* sim may be NULL: last create error */

/* gymtorch.wrap_tensor(gym.acquire_*_tensor(sim)): device pointer + shape of a sim-owned
 * buffer.  shape has `*ndim` valid entries. */
int bez_sim_get_tensor(BezSim* sim, int which, void** dev_ptr, int64_t shape[3], int* ndim, int* dtype);

/* gym.refresh_{actor_root_state,dof_state,rigid_body_state,net_contact_force}_tensor
 * (kick_env.py:750-753): materialise the Isaac-layout tensor from the SoA state. */
int bez_sim_refresh_tensor(BezSim* sim, int which, void* stream);

/* gym.set_actor_root_state_tensor_indexed (kick_env.py:831-837): teleport the actors listed in
 * actor_ids_dev (sim-domain actor index = env*2 + {0 robot, 1 ball}) to the rows of the full
 * (N*2,13) tensor root_states_dev. */
int bez_sim_set_actor_root_state_tensor_indexed(BezSim* sim, const float* root_states_dev,
                                                const int32_t* actor_ids_dev, int32_t n, void* stream);
/* gym.set_dof_state_tensor_indexed (kick_env.py:844-847); actor ids as above (robot actors). */
int bez_sim_set_dof_state_tensor_indexed(BezSim* sim, const float* dof_state_dev,
                                         const int32_t* actor_ids_dev, int32_t n, void* stream);
/* gym.set_dof_position_target_tensor (kick_env.py:419): (N,18) targets, used as given. */
int bez_sim_set_dof_position_target_tensor(BezSim* sim, const float* targets_dev, void* stream);
/* gym.set_dof_position_target_tensor_indexed (kick_env.py:839-842). */
int bez_sim_set_dof_position_target_tensor_indexed(BezSim* sim, const float* targets_dev,
                                                   const int32_t* actor_ids_dev, int32_t n, void* stream);
/* Writes the (N*22,3) net-contact-force tensor into the sim (test hook: the reference's feet
 * logic reads AND mutates this tensor, kick_env.py:987-990). */
int bez_sim_set_net_contact_force_tensor(BezSim* sim, const float* forces_dev, void* stream);

/* VecTask.step's clamp + KickEnv.pre_physics_step (vec_task.py:317, kick_env.py:410-419):
 * a = clamp(actions, +-clip); a[:,0:2] = 0; target = clamp(a + default_dof_pos, lower, upper). */
int bez_sim_pre_physics(BezSim* sim, const float* actions_dev, void* stream);
/* gym.simulate (vec_task.py:322-324): one control step dt = `substeps` substeps of articulated
 * dynamics (ABA, implicit PD drives, contact) for every env. */
int bez_sim_simulate(BezSim* sim, void* stream);
/* timeout fill + KickEnv.post_physics_step (vec_task.py:331-335, kick_env.py:426-438):
 * timeout = progress >= max_len-1; progress += 1; reset_idx(envs with reset_buf != 0);
 * observations; reward + reset flags. */
int bez_sim_post_physics(BezSim* sim, void* stream);
/* The three calls above fused into ONE kernel launch: the whole of VecTask.step
 * (vec_task.py:303-349) for all envs.  This is the hot path. */
int bez_sim_step(BezSim* sim, const float* actions_dev, void* stream);
/* bez_sim_step for n_steps consecutive control steps, actions_dev = (n_steps, N, 18); the
 * rollout inner loop without a host round trip per step (random-action benchmark). */
int bez_sim_step_many(BezSim* sim, const float* actions_dev, int32_t n_steps, void* stream);

/* KickEnv.reset_idx (kick_env.py:779-850) for the env ids in env_ids_dev. */
int bez_sim_reset_indexed(BezSim* sim, const int32_t* env_ids_dev, int32_t n, void* stream);

/* Domain randomisation parameters (vec_task.py:505-725 -> per-env arrays read by the kernel).
 * values_dev: (N, count) fp32 where count is fixed per param; NULL restores the default. */
enum BezEnvParam {
  BEZ_PARAM_FRICTION = 0,   /* (N,1)  ground/ball friction coefficient      bez_kick.yaml:179-186 */
  BEZ_PARAM_KP_SCALE = 1,   /* (N,18) stiffness scaling                     bez_kick.yaml:200-205 */
  BEZ_PARAM_KD_SCALE = 2,   /* (N,18) damping scaling                       bez_kick.yaml:194-199 */
  BEZ_PARAM_MASS_SCALE = 3, /* (N,19) link mass scaling (setup only)        bez_kick.yaml:170-177 */
  BEZ_PARAM_GRAVITY = 4,    /* (N,3)  gravity vector                        bez_kick.yaml:162-167 */
  BEZ_PARAM_DOF_LOWER = 5,  /* (N,18) physical lower joint limits           bez_kick.yaml:206-212 */
  BEZ_PARAM_DOF_UPPER = 6,  /* (N,18) physical upper joint limits           bez_kick.yaml:213-219 */
  BEZ_PARAM_COUNT = 7
};
int bez_sim_set_env_params(BezSim* sim, int param, const float* values_dev, void* stream);
/* copies the current (N, count) array of `param` into out_dev (defaults if the parameter was never set) */
int bez_sim_get_env_params(BezSim* sim, int param, float* out_dev, void* stream);

/* Device-side domain randomisation: VecTask.apply_randomizations (vec_task.py:505-725, called from reset_idx, kick_env.py:781-782)
 * without the host.  One entry of cfg/task/bez_kick.yaml:151-219 per BezDrRange: a, b = `range`; distribution and operation are
 * fixed per parameter as that file has them (uniform scaling: friction, stiffness, damping; gaussian additive, b = the number
 * numpy uses as the std: lower, upper, gravity, observations, actions); schedule_steps > 0 = `schedule: linear` (the range is
 * interpolated from "no randomisation" by min(frame, schedule_steps) / schedule_steps, vec_task.py:560-566 and gymutil [ext]).
 * Once set, every call that contains the post-physics first runs one small kernel that does what reset_idx's call does:
 * randomize_buf += 1; an env with reset_buf != 0 and randomize_buf >= frequency draws new friction / Kp / Kd / joint limits from
 * the Philox stream keyed by (seed, GLOBAL env id, episode, parameter) and clears its randomize_buf; if any env resets and
 * `frequency` frames have passed since the last time, gravity (one draw for the whole sim) and the noise parameters are
 * refreshed.  Nothing syncs with the host: the step stays HIP-graph capturable.  The first call randomises every env at
 * frame 0 (first_randomization, vec_task.py:521-523).  NULL switches it off (arrays keep their values). */
typedef struct BezDrRange {
  float a, b;
  int32_t enabled;
  int32_t schedule_steps;
} BezDrRange;
typedef struct BezDrConfig {
  int32_t frequency;        /* randomization_params.frequency                     bez_kick.yaml:153 */
  int32_t friction_buckets; /* rigid_shape_properties.friction.num_buckets        bez_kick.yaml:180 */
  BezDrRange friction;      /* bez_kick.yaml:179-186 */
  BezDrRange stiffness;     /* bez_kick.yaml:200-205 */
  BezDrRange damping;       /* bez_kick.yaml:194-199 */
  BezDrRange lower, upper;  /* bez_kick.yaml:206-219 */
  BezDrRange gravity;       /* bez_kick.yaml:162-167 */
  BezDrRange observations;  /* bez_kick.yaml:154-157 */
  BezDrRange actions;       /* bez_kick.yaml:158-161 */
} BezDrConfig;
int bez_sim_set_randomization(BezSim* sim, const BezDrConfig* dr, void* stream);
/* The randomisation kernel of the COMING control step (what the next call containing the post-physics would launch first), now, on
 * `stream`; that call then skips its own.  Lets a caller overlap it with whatever else precedes the step (a policy forward pass) on
 * another stream -- which must be joined before the step.  The previous step must be complete on a stream `stream` is ordered behind.
 * The action noise (bez_sim_add_dr_noise(which = 1), BezActionNoiseSource) reads a SNAPSHOT the previous step left, not the state this
 * kernel updates, so it may run concurrently with it. */
int bez_sim_dr_prelaunch(BezSim* sim, void* stream);
/* The same work as an opaque argument block for a launch of the caller's that executes it ITSELF, as one extra workgroup beside its own
 * (bez_ppo_policy_rollout_step's dr_step argument): fills blob[BEZ_DR_STEP_BYTES] for the COMING control step, which then skips its own
 * randomisation kernel -- the caller must run that launch before the step, on the step's stream.  Nothing in the block changes from step
 * to step (all clocks live in device memory), so a captured graph replays it.  Returns the bytes used, 0 without a randomisation. */
#define BEZ_DR_STEP_BYTES 512
int bez_sim_dr_step_args(BezSim* sim, void* blob, int32_t blob_bytes);
/* Takes back a hand-out of bez_sim_dr_prelaunch / bez_sim_dr_step_args whose consumer did NOT run (its launch failed or was skipped): the
 * coming control step launches its own randomisation kernel again.  Also implied by bez_sim_set_randomization and bez_sim_seed.  Only for
 * the not-run case: after a consumer that did run, the step would randomise twice. */
int bez_sim_dr_cancel(BezSim* sim);
/* For a consumer that adds the action noise of vec_task.py:586-592 itself (e.g. bez_ppo_policy_rollout_step): *snap_dev -> device struct
 * {float mean, std; uint32_t frame_lo, frame_hi} kept current by the step kernels; noise of element i of the flat (N, 18) action tensor =
 * mean + std * z, z = word (i & 3) of the Philox4x32-10 block with counter (key lo, key hi, frame lo, 0x4e4f4953 + 1 + (frame hi << 8)),
 * key = env_id_offset * 64 + (i >> 2), Philox key = seed, Box-Muller pairs (words 0,1 -> z0 = r cos, z1 = r sin; words 2,3 -> z2, z3) --
 * bit for bit what bez_sim_add_dr_noise(which = 1) adds.  Returns 1 if an action noise is configured, 0 if not, < 0 on error. */
int bez_sim_action_noise_source(BezSim* sim, const void** snap_dev, uint64_t* seed, int64_t* env_id_offset);
/* The noise lambdas of vec_task.py:544-618 in one launch: y[i] = x[i] + mean + std * N(0,1) for n floats (y_dev may be x_dev: in
 * place), mean / std = the current entries of BEZ_TENSOR_DR_NOISE for `which` (0 = observations, 1 = actions), normals from
 * Philox keyed by (seed, frame, which, i / 4) -- one call per control step and kind. */
int bez_sim_add_dr_noise(BezSim* sim, const float* x_dev, float* y_dev, int64_t n, int32_t which, void* stream);

/* Test hooks (state injection for the parity tests; no reference counterpart). */
int bez_sim_set_prev_lin_vel_tensor(BezSim* sim, const float* prev_dev, void* stream); /* (N,3) */
int bez_sim_set_flags(BezSim* sim, uint32_t flags);  /* the asset bits (BEZ_FLAG_CLEATS, BEZ_FLAG_BOX_ASSET) keep their creation value */
/* compute_observations + compute_reward on the current state, without timeout/progress/reset bookkeeping */
int bez_sim_observe_reward(BezSim* sim, void* stream);
/* writes the per-env goal (N,2) of bez_walk / bez_orient */
int bez_sim_set_goal_tensor(BezSim* sim, const float* goal_dev, void* stream);
/* number of compute_observations passes already done: only pass 0 differences against prev = zeros (Q1) */
int bez_sim_set_obs_calls(BezSim* sim, int64_t calls);

/* Re-keys the reset-noise stream (utils/utils.py:45-70 set_seed). */
int bez_sim_seed(BezSim* sim, uint64_t seed);

/* Measurement utility: launches a known-size dword-per-lane read (write=0) or write (write=1) of n_floats floats of
 * buf_dev, to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE for this library's access shape. */
int bez_sim_calibrate(void* buf_dev, uint64_t n_floats, int32_t write, void* stream);

/* Average device time [ms] of the fused step kernel over `n_steps` launches on `stream`,
 * measured with HIP events recorded on that same stream (bench.py roofline leg). */
int bez_sim_time_steps(BezSim* sim, const float* actions_dev, int32_t n_steps, void* stream, float* avg_ms);

/* ------------------------------------------------------------------------------------------------------------------
 * PPO glue kernels (bez_ppo.hip).  The reference trains through rl_games' a2c_continuous (call sites train.py:89-113,
 * hyper-parameters cfg/train/bez_kickPPO.yaml); its MLP stays in PyTorch-ROCm, these entry points replace the ~250 tiny
 * elementwise / reduction launches per minibatch step around it.  All pointers are device memory, fp32 unless noted. */

/* The signatures below change between rounds (round 3: scratch buffers of the fixed-order reductions, plan / run split of the weight
 * gradients, the optimiser tail's bookkeeping): a binding checks this number once after dlopen. */
#define BEZ_PPO_ABI_VERSION 9
int32_t bez_ppo_abi_version(void);

/* RunningMeanStd (normalize_input / normalize_value, bez_kickPPO.yaml:51-52): moments[0:D] = column sums, [D:2D] = sums of
 * squares, [2D] = rows, in fp64 (the caller may all-reduce them across ranks before applying).  scratch_dev: NULL = fp64 atomics (the
 * last bits vary from run to run); otherwise 1 + 1024 * 2 * cols doubles: per-workgroup partials, added in workgroup order by a second
 * small launch -- bit-reproducible. */
int bez_ppo_rms_moments(const float* x_dev, int64_t rows, int32_t cols, double* moments_dev, double* scratch_dev, void* stream);
int bez_ppo_rms_apply(const double* moments_dev, int32_t cols, double* mean_dev, double* var_dev, double* count_dev, void* stream);
/* y = clamp((x - mean) / sqrt(var + eps), -5, 5); y_dev is fp32 or (out_f16 != 0) fp16 */
int bez_ppo_rms_normalize(const float* x_dev, int64_t rows, int32_t cols, const double* mean_dev, const double* var_dev, float eps,
                          void* y_dev, int32_t out_f16, void* stream);
/* rollout: actions = mu + exp(logstd) * noise, neglogp(actions), env_actions = clamp(actions, -1, 1), sigma broadcast */
int bez_ppo_sample(const float* mu_dev, const float* logstd_dev, const float* noise_dev, int64_t n, int32_t num_actions,
                   float* actions_dev, float* env_actions_dev, float* neglogp_dev, float* sigma_dev, void* stream);
/* One rollout step between the policy forward and the env step (rl_games a2c_common.play_steps [ext] via train.py:89-113):
 * network outputs mu (n, num_actions) / value (n) as fp16 (inputs_f16 != 0) or fp32; the value is de-normalised with the running
 * value statistics (NULL = none); rows of the rollout buffers obs / dones / mu / value are written; actions are sampled as
 * bez_ppo_sample does. */
int bez_ppo_rollout_pre(const void* mu_dev, const void* value_dev, int32_t inputs_f16, const float* logstd_dev, const float* noise_dev, const float* obs_dev,
                        const float* dones_dev, const double* value_mean_dev, const double* value_var_dev, float value_eps, int64_t n, int32_t num_actions,
                        int32_t num_obs, float* mb_obs_dev, float* mb_dones_dev, float* mb_mu_dev, float* mb_val_dev, float* actions_dev,
                        float* env_actions_dev, float* neglogp_dev, float* sigma_dev, void* stream);

/* rollout bookkeeping of one env step: shaped = rew * reward_scale (+ gamma * value * time_out, value_bootstrap),
 * dones as float, running episode return / length, ep_stats[3] += (finished, sum of returns, sum of lengths) in fp64 */
int bez_ppo_rollout_post(const float* rew_dev, const int64_t* dones_dev, const int64_t* timeouts_dev, const float* values_dev, int64_t n,
                         float reward_scale, float gamma, int32_t value_bootstrap, float* shaped_dev, float* dones_f_dev,
                         float* cur_rew_dev, float* cur_len_dev, double* ep_stats_dev, void* stream);
/* ABI 8: the same launch with ONE EXTRA workgroup that adds the nslots per-workgroup slots of BezPpoRolloutPost.ep_parts (4 doubles each) to
 * ep_stats and clears them -- the last bookkeeping launch of a rollout whose policy launches filled the slots. */
int bez_ppo_rollout_post_fold(const float* rew_dev, const int64_t* dones_dev, const int64_t* timeouts_dev, const float* values_dev, int64_t n,
                              float reward_scale, float gamma, int32_t value_bootstrap, float* shaped_dev, float* dones_f_dev, float* cur_rew_dev,
                              float* cur_len_dev, double* ep_stats_dev, double* ep_parts_dev, int32_t nslots, void* stream);
/* PPO minibatch loss (clipped surrogate, clipped value loss, entropy, bounds loss) AND its gradient w.r.t. the network
 * outputs mu (B,A), value (B) and the log-std parameter (A), multiplied by *loss_scale_dev (GradScaler; NULL = 1).
 * stats_dev[5] = sums of a_loss, c_loss, b_loss, KL(current || old), entropy over the minibatch.
 * clip_value: bit 0 = clipped value loss; bit 1 = ACCUMULATE into grad_logstd_dev instead of clearing it first; bit 2 = stats_dev was
 * zeroed by the caller (e.g. as part of the flat gradient buffer's one clear per step); bit 3 = after the KL is taken, WRITE the
 * current mu and sigma = exp(logstd) over old_mu_dev / old_sigma_dev (rl_games' PPODataset.update_mu_sigma [ext]: from the second
 * mini-epoch on the KL is measured against the previous pass over the minibatch).
 * scratch_dev: NULL = the A + 5 sums are float atomics (last bits differ from run to run); otherwise 2 + ceil(batch / 64) * (A + 5)
 * floats: every workgroup stores its partials there and a second one-wave launch adds them in a fixed order -- bit-reproducible. */
int bez_ppo_loss(const float* mu_dev, const float* logstd_dev, const float* value_dev, const float* actions_dev, const float* old_logp_dev,
                 const float* adv_dev, const float* old_value_dev, const float* returns_dev, const float* old_mu_dev,
                 const float* old_sigma_dev, int64_t batch, int32_t num_actions, float e_clip, float critic_coef, float entropy_coef,
                 float bounds_coef, int32_t clip_value, const float* loss_scale_dev, float* grad_mu_dev, float* grad_value_dev,
                 float* grad_logstd_dev, float* stats_dev, float* scratch_dev, void* stream);

/* The rollout's policy forward pass (rl_games get_action_values [ext] via train.py:89-113) in one launch: observation normaliser
 * (NULL mean = none), num_hidden Linear + ELU layers, the mu head (num_actions <= 31) and the value head, on fp16 weights /
 * biases in torch layout ((out, in) row-major; hidden_* are HOST arrays of num_hidden device pointers / widths <= 416), fp32
 * accumulation, fp16 rounding after every Linear and ELU as torch's fp16 path does.  Outputs fp32: mu (n, num_actions), value (n). */
int bez_ppo_policy_forward(const float* obs_dev, int64_t n, int32_t num_obs, const double* obs_mean_dev, const double* obs_var_dev, float obs_eps,
                           int32_t num_hidden, const void* const* hidden_w_f16_dev, const void* const* hidden_b_f16_dev, const int32_t* hidden_width,
                           const void* mu_w_f16_dev, const void* mu_b_f16_dev, int32_t num_actions, const void* value_w_f16_dev,
                           const void* value_b_f16_dev, float* mu_dev, float* value_dev, int32_t weights_packed, void* stream);

/* The fused rollout step: bez_ppo_policy_forward followed, inside the same launch, by bez_ppo_rollout_pre's work (rollout-buffer
 * rows of obs / dones / mu / de-normalised value, a = mu + exp(logstd) * noise, neglogp, the clamped env action).  mu / value
 * never visit HBM in fp16; replaces a2c_common.py:1105-1135's per-step get_action_values + buffer updates.
 * prev_post (NULL = none; ABI 4): the bookkeeping of the env step BEFORE this one rides in the same launch -- exactly
 * bez_ppo_rollout_post(rew, reset, timeouts, prev_values, ...) on the env's still-unchanged reward / reset / time-out buffers, with
 * dones_f also written to this step's mb_dones row (dones_dev is then not read): a rollout step is two launches (policy, env). */
typedef struct BezPpoRolloutPost {
  const float* rew; const int64_t* reset; const int64_t* timeouts; const float* prev_values;
  float reward_scale, gamma; int32_t bootstrap;
  float* shaped; float* dones_f; float* cur_rew; float* cur_len; double* ep_stats;
  /* ABI 8 (NULL = the launch adds to ep_stats with atomics, as before): 4 doubles per workgroup of the launch (ceil(n / 32) of them at most), each
   * workgroup ADDS its finished episodes' (count, return sum, length sum) to its own slot -- no atomics: 128-256 workgroups adding to one cache line
   * cost the launch 1.7-3.2 us.  The caller adds the slots to ep_stats and clears them once per rollout. */
  double* ep_parts;
} BezPpoRolloutPost;
/* action_noise (NULL = none; ABI 4): the env's action-noise lambda of the domain randomisation inside this launch -- env_actions_dev receives
 * clamp(a, -1, 1) + noise, bit for bit what bez_sim_add_dr_noise(which = 1) would add to the clamped actions (the three fields are what
 * bez_sim_action_noise_source returns; the env must then not add it again). */
typedef struct BezPpoActionNoise { const void* snap_dev; uint64_t seed; int64_t env_id_offset; } BezPpoActionNoise;
/* layout (NULL = contiguous rows; ABI 4): row strides, in floats, of the rollout rows this launch writes -- mb_obs (>= num_obs); mb_mu / actions /
 * sigma (>= num_actions); neglogp (>= 1).  Lets the caller point them INTO its env-major dataset tensors (row of env e at step n = e * H + n:
 * strides H * width), so that the dataset needs no transposing copies after the rollout. */
typedef struct BezPpoRolloutLayout { int64_t obs_row_stride, act_row_stride, scalar_stride; } BezPpoRolloutLayout;
/* dr_step (NULL = none; ABI 4): the block bez_sim_dr_step_args filled -- the coming env step's randomisation runs as ONE EXTRA workgroup of this
 * launch (it touches nothing the forward pass reads: the action noise comes from the snapshot), instead of a launch of its own in front of
 * the step. */
int bez_ppo_policy_rollout_step(const float* obs_dev, int64_t n, int32_t num_obs, const double* obs_mean_dev, const double* obs_var_dev, float obs_eps,
                                int32_t num_hidden, const void* const* hidden_w_f16_dev, const void* const* hidden_b_f16_dev, const int32_t* hidden_width,
                                const void* mu_w_f16_dev, const void* mu_b_f16_dev, int32_t num_actions, const void* value_w_f16_dev,
                                const void* value_b_f16_dev, const float* logstd_dev, const float* noise_dev, const float* dones_dev,
                                const double* value_mean_dev, const double* value_var_dev, float value_eps, float* mb_obs_dev, float* mb_dones_dev,
                                float* mb_mu_dev, float* mb_val_dev, float* actions_dev, float* env_actions_dev, float* neglogp_dev, float* sigma_dev,
                                int32_t weights_packed, const BezPpoRolloutPost* prev_post, const BezPpoActionNoise* action_noise,
                                const void* dr_step, const BezPpoRolloutLayout* layout, void* stream);

/* The forward half of a PPO minibatch step (a2c_common.py calc_gradients: model(batch) under autocast): bez_ppo_policy_forward
 * that also keeps what the backward pass needs -- x0 (n, num_obs) fp16 = the normalised, clamped input of the first Linear, and
 * act[i] (n, hidden_width[i]) fp16 = the output of ELU i.  num_obs and the widths must be even. */
int bez_ppo_policy_forward_train(const float* obs_dev, int64_t n, int32_t num_obs, const double* obs_mean_dev, const double* obs_var_dev, float obs_eps,
                                 int32_t num_hidden, const void* const* hidden_w_f16_dev, const void* const* hidden_b_f16_dev, const int32_t* hidden_width,
                                 const void* mu_w_f16_dev, const void* mu_b_f16_dev, int32_t num_actions, const void* value_w_f16_dev,
                                 const void* value_b_f16_dev, void* x0_f16_dev, void* const* act_f16_dev, float* mu_dev, float* value_dev, int32_t weights_packed, void* stream);

/* Gradient reductions of explicit-fp16 linear layers into the fp32 master gradient: the sum over `splits` split-K partial
 * products ([splits][n] fp16) and the bias gradient = column sums of dY ((rows, cols) fp16).  accumulate != 0 adds to out_dev. */
int bez_ppo_wgrad_sum(const void* partials_f16_dev, int32_t splits, int64_t n, float* out_dev, int32_t accumulate, void* stream);
int bez_ppo_colsum_f16(const void* y_f16_dev, int64_t rows, int32_t cols, float* out_dev, int32_t accumulate, void* stream);
/* The weight gradients of `nlayers` (<= 8) Linear layers in one launch pair (csrc/bez_wgrad.hip): dW_L (+)= dY_L^T X_L with dY_L
 * (rows, out_L) and X_L (rows, in_L) fp16 row-major, dW_L (out_L, in_L) fp32.  A split-K MFMA kernel over the output blocks of
 * all layers -- each split along the rows in proportion to the bytes it streams, ~250 workgroups in all -- writes fp32 partial
 * blocks into partial_dev (room for nsplit * sum(out_L * in_L) floats; it uses what the balance needs), a second kernel adds a
 * block's splits in fixed order (deterministic).  Replaces torch.bmm + bez_ppo_wgrad_sum of a2c_common.py's backward [ext].
 * bez_ppo_wgrad_plan lays the work out ONCE for a set of tensors into plan_host (BEZ_PPO_WGRAD_PLAN_BYTES bytes of host memory;
 * rows % 64 must be 0; -3 = shapes the kernel does not take, the caller keeps its GEMM path); the caller copies the plan to
 * device memory and calls bez_ppo_wgrad_run(plan_host, plan_dev, accumulate, stream) per step. */
#define BEZ_PPO_WGRAD_PLAN_BYTES 3072
int bez_ppo_wgrad_plan(const void* const* dy_f16_dev, const void* const* x_f16_dev, const int32_t* out_features, const int32_t* in_features,
                       float* const* dw_dev, int32_t nlayers, int64_t rows, int32_t nsplit, float* partial_dev, void* plan_host);
int bez_ppo_wgrad_run(const void* plan_host, const void* plan_dev, int32_t accumulate, void* stream);
/* Every second-stage reduction of a minibatch step's gradient in ONE launch (ABI 4): the split-K images of bez_ppo_wgrad_run called
 * with accumulate = 2 ("partial images only"), the per-workgroup bias column sums of bez_ppo_policy_backward called with bit 1 of
 * weights_packed set ("column sums only"; same partial_dev, rows, widths and gradient pointers as there) and the per-workgroup sums of
 * bez_ppo_loss called with bit 4 of clip_value set ("partials only"; same scratch_dev, batch = loss_rows, grad_logstd_dev, stats_dev).
 * Fixed-order sums, one writer per output element: with accumulate = 0 the launch WRITES all weight / bias / log-sigma gradients and
 * the five statistics, so nothing has to be cleared in front of a step. */
int bez_ppo_grad_reduce_all(const void* plan_host, const void* plan_dev, const float* bias_partial_dev, int64_t rows, int32_t num_hidden,
                            const int32_t* hidden_width, int32_t num_actions, float* const* bias_grad_dev, float* mu_bias_grad_dev,
                            float* value_bias_grad_dev, const float* loss_scratch_dev, int64_t loss_rows, float* grad_logstd_dev, float* stats_dev,
                            int32_t accumulate, float* norm_parts_dev, void* stream);
/* norm_parts_dev (NULL = none; used with accumulate = 0 only): receives per workgroup (sum of squares, count of non-finite values) of the
 * gradient elements that workgroup wrote -- bez_ppo_grad_reduce_blocks(...) pairs of floats -- for bez_ppo_adam_step (BezPpoAdamExtra). */
int bez_ppo_grad_reduce_blocks(const void* plan_host, int32_t num_hidden, const int32_t* hidden_width, int32_t num_actions);
/* ELU (alpha 1) backward fused with the bias gradient: gz = gy * elu'(y) from the layer's ELU OUTPUT y, all (rows, cols) fp16;
 * the column sums of gz go to bias_grad_dev (fp32, cols). */
int bez_ppo_elu_bwd_colsum_f16(const void* gy_f16_dev, const void* y_f16_dev, void* gz_f16_dev, int64_t rows, int32_t cols, float* bias_grad_dev,
                               int32_t accumulate, void* stream);

/* The input-gradient half of a PPO minibatch's backward pass in one launch (torch autograd through the actor-critic MLP of
 * a2c_common.py calc_gradients): from d loss / d mu (n, A) and d loss / d value (n, 1) to gz[i] (n, hidden_width[i]) fp16 = d loss / d
 * (pre-activation of hidden layer i), using the ELU outputs act[i] kept by bez_ppo_policy_forward_train.  Also written: the fp16 casts of
 * the two loss gradients (operands of the head weight-gradient GEMMs); ADDED by atomics: the bias gradients of every hidden layer and of the
 * two heads.  Weight operands are transposed fp16 copies: wt[i] = W_i^T (hidden_width[i-1], hidden_width[i]) for i >= 1 (wt[0] unused) and
 * heads_t (hidden_width[last], 32) = [Wmu^T | Wvalue^T | 0]; bez_ppo_scatter_f16 refreshes them from a flat fp16 copy through an index map.
 * partial_dev: scratch of ceil(n / 64) * (sum(hidden_width) + 32) floats (per-workgroup column sums of the hidden layers and of the
 * two heads, added in fixed order by a second small launch: every bias gradient is bit-reproducible).
 * Widths even, 32 <= width <= 416. */
int bez_ppo_policy_backward(const float* grad_mu_dev, const float* grad_value_dev, int64_t n, int32_t num_hidden, const int32_t* hidden_width,
                            int32_t num_actions, const void* const* act_f16_dev, const void* const* wt_f16_dev, const void* heads_t_f16_dev,
                            void* const* gz_f16_dev, void* grad_mu_f16_dev, void* grad_value_f16_dev, float* const* bias_grad_dev,
                            float* mu_bias_grad_dev, float* value_bias_grad_dev, float* partial_dev, int32_t weights_packed, void* stream);
int bez_ppo_scatter_f16(const void* src_f16_dev, const int32_t* map_dev, int64_t n, void* dst_f16_dev, void* stream); /* dst[map[i]] = src[i], map[i] >= 0 */
int bez_ppo_scatter2_f16(const void* src_f16_dev, const int32_t* map_a_dev, const int32_t* map_b_dev, int64_t n, void* dst_f16_dev, void* stream);

/* weights_packed (last int argument of the four bez_ppo_policy_* functions above): 0 = the weight pointers are the row-major (out, in) fp16
 * matrices of nn.Linear (transposed ones for the backward function, as described there); 1 = they are FRAGMENT-MAJOR copies: for a Linear
 * (out, in), block nb = n / 32 and k-step ks = k / 16 form one 1 KB chunk at halfs ((nb * ceil(in / 16) + ks) * 512), in which element
 * (n, k) sits at ((k / 8 % 2) * 32 + n % 32) * 8 + k % 8, zero-padded beyond out / in -- a wave's MFMA B fragment as one coalesced load.
 * Forward: hidden_w[i] packs W_i, mu_w packs the (num_actions + 1, width_last) matrix [W_mu; W_value] (value_w unused).  Backward: wt[i]
 * packs W_i^T as a Linear (out = hidden_width[i-1], in = hidden_width[i]), heads_t packs [W_mu; W_value]^T as (out = width_last, in = 32).
 * bez_ppo_scatter2_f16 refreshes both sets from one flat fp16 copy of the parameters (two index maps) in one launch. */

/* rl_games AdaptiveScheduler.update (the `lr_schedule: adaptive` rule of bez_kickPPO.yaml) on device scalars: *lr /= 1.5 when *kl > 2 *
 * kl_threshold (floor min_lr), then *lr *= 1.5 when *kl < kl_threshold / 2 (cap max_lr).  No host round trip. */
int bez_ppo_adaptive_lr(float* lr_dev, const float* kl_dev, float kl_threshold, float min_lr, float max_lr, void* stream);

/* GAE (rl_games a2c_common.py discount_values, called from play_steps): advantages (H,N) from rewards / values (H,N), the done
 * flags recorded BEFORE each step (H,N), the current done flags (N) and the bootstrap values (N); returns_dev (optional) = advantages +
 * values.  One thread per env, the reference's operation order.  value_mean_dev != NULL: last_values_dev holds the network's NORMALISED
 * value outputs and the kernel de-normalises them first (RunningMeanStd(unnorm=True): clamp +-5, * sqrt(var + eps), + mean). */
int bez_ppo_gae(const float* rewards_dev, const float* values_dev, const float* mb_dones_dev, const float* dones_dev, const float* last_values_dev,
                int32_t horizon, int64_t num_envs, float gamma, float tau, float* advantages_dev, float* returns_dev, const double* value_mean_dev,
                const double* value_var_dev, float value_eps, void* stream);

/* bez_ppo_loss + bez_ppo_policy_backward of one minibatch as ONE launch: every 64-row tile forms its loss terms and d loss / d mu,
 * d loss / d value in front of the backward chain (the same code, the same bits), which reads them from on-chip memory.  The loss's per-
 * workgroup partial sums go to loss->scratch_dev (bez_ppo_loss's scratch, bit 4 semantics: bez_ppo_grad_reduce_all writes d loss / d
 * log-sigma and the statistics); weights_packed_flags must carry bits 0 and 1 (fragment-major weights, column sums only).  clip_value: bits 0
 * and 3 of bez_ppo_loss.  -3: a shape the fused kernel is not instantiated for (it exists for num_actions == 18 and hidden layers that fit
 * the two-workgroups-per-CU tiles) -- the caller keeps the two separate calls. */
typedef struct BezPpoLossOperands {
  const float* mu_dev; const float* logstd_dev; const float* value_dev; const float* actions_dev; const float* old_logp_dev; const float* adv_dev;
  const float* old_value_dev; const float* returns_dev; const float* old_mu_dev; const float* old_sigma_dev;
  float e_clip, critic_coef, entropy_coef, bounds_coef;
  int32_t clip_value;
  const float* loss_scale_dev;
  float* scratch_dev;
} BezPpoLossOperands;
int bez_ppo_policy_backward_with_loss(const BezPpoLossOperands* loss, int64_t n, int32_t num_hidden, const int32_t* hidden_width, int32_t num_actions,
                                      const void* const* act_f16_dev, const void* const* wt_f16_dev, const void* heads_t_f16_dev, void* const* gz_f16_dev,
                                      void* grad_mu_f16_dev, void* grad_value_f16_dev, float* const* bias_grad_dev, float* mu_bias_grad_dev,
                                      float* value_bias_grad_dev, float* partial_dev, int32_t weights_packed_flags, void* stream);

/* The epoch's dataset preparation between GAE and the first minibatch (rl_games a2c_continuous.py prepare_dataset [ext], called from
 * train_epoch via train.py:89-113, + the per-minibatch moments RunningMeanStd absorbs at every training forward) in four launches:
 *   - obs_moments_dev[i] (2 num_obs + 1 doubles: column sums, sums of squares, rows) of minibatch i's rows of obs_dev
 *     ((num_minibatches * minibatch_rows, num_obs) fp32), and value_moments_dev / return_moments_dev (3 doubles) of values_dev / returns_dev
 *     ((horizon, num_envs) fp32, the rollout's layout);
 *   - with value_mean_dev != NULL, RunningMeanStd.forward in train mode on the values, then on the returns: update with the values' moments,
 *     normalise the values (clamp +-5), update with the returns' moments, normalise the returns; the statistics (mean / var / count
 *     doubles) are updated in place;
 *   - old_values_dev, ds_returns_dev, advantages_dev (horizon * num_envs fp32, ENV-major: row e * horizon + t, swap_and_flatten01's order):
 *     the normalised values / returns and advantage = return - value, normalised as (adv - mean) / (std + 1e-8) (torch's unbiased std)
 *     when normalize_advantage != 0.
 * Every sum is a fixed-order two-stage fp64 sum (bit-reproducible).  scratch_dev: (num_minibatches + 2) * 256 * 128 + 2 * ceil(horizon *
 * num_envs / 256) doubles at most.  -3: horizon * num_envs is not a multiple of 64 (the caller keeps its separate launches). */
int bez_ppo_dataset_prep(const float* obs_dev, int64_t minibatch_rows, int32_t num_minibatches, int32_t num_obs, double* obs_moments_dev,
                         const float* values_dev, const float* returns_dev, int32_t horizon, int64_t num_envs, double* value_mean_dev, double* value_var_dev,
                         double* value_count_dev, float value_eps, double* value_moments_dev, double* return_moments_dev, float* old_values_dev,
                         float* ds_returns_dev, float* advantages_dev, int32_t normalize_advantage, double* scratch_dev, int64_t scratch_doubles, void* stream);

/* The same work in STAGES, for the data-parallel loop, whose two per-epoch collectives sit between them -- every moment above is
 * a plain sum (sum x, sum x^2, count), so an all-reduce of the local moments yields the global batch's:
 *   stage 1  the observation / value / return moments of THIS rank's rows                 (2 launches)   -> all-reduce of the moment buffers
 *   stage 2  value-normaliser statistics from the (now global) moments, normalised values / returns, advantages, and adv_sums_dev[0..2] =
 *            (sum adv, sum adv^2, count) of this rank's rows                              (2 launches)   -> all-reduce of adv_sums_dev
 *   stage 4  advantage normalisation with the (now global) adv_sums_dev; commits the value normaliser's statistics       (1 launch)
 * `stages` is a bit mask; 1 | 2 | 4 with adv_sums_dev == NULL is bez_ppo_dataset_prep.  The paths differ by the collectives only. */
int bez_ppo_dataset_prep_staged(int32_t stages, const float* obs_dev, int64_t minibatch_rows, int32_t num_minibatches, int32_t num_obs,
                                double* obs_moments_dev, const float* values_dev, const float* returns_dev, int32_t horizon, int64_t num_envs,
                                double* value_mean_dev, double* value_var_dev, double* value_count_dev, float value_eps, double* value_moments_dev,
                                double* return_moments_dev, float* old_values_dev, float* ds_returns_dev, float* advantages_dev,
                                int32_t normalize_advantage, double* adv_sums_dev, double* scratch_dev, int64_t scratch_doubles, void* stream);

/* The backward inputs of the two heads in one pass over the loss gradients (torch.autocast's cast nodes + the bias-gradient sums of
 * nn.Linear's backward): fp16 copies of d loss / d mu (rows, num_actions) and d loss / d value (rows, 1), and the column sums of those
 * fp16 values ADDED to the fp32 bias gradients of the mu and value heads. */
int bez_ppo_head_grads_f16(const float* grad_mu_dev, const float* grad_value_dev, int64_t rows, int32_t num_actions, void* grad_mu_f16_dev,
                           void* grad_value_f16_dev, float* mu_bias_grad_dev, float* value_bias_grad_dev, void* stream);

/* The optimiser tail of one minibatch step on flat fp32 buffers of n elements (replaces rl_games' scaler.unscale_ +
 * clip_grad_norm_ + scaler.step(Adam) + scaler.update, a2c_common.py [ext] via train.py:89-113): the gradient is divided by
 * *scale_dev (NULL = no loss scaling), clipped to max_norm (<= 0: no clipping), applied with torch's Adam formula; a non-finite
 * gradient skips the step and backs the scale off, growth_interval clean steps grow it.  steps_dev[nsteps] are the per-tensor
 * step counters of the optimiser state (all equal).  ONE launch (ABI 4; three before): every workgroup forms the squared norm of
 * the whole gradient itself, in the same fixed order (no float atomics: the clip coefficient, and with it every weight, is
 * bit-identical run to run and rank to rank), updates its slice, and the workgroup that draws the last ticket commits the step
 * counters / loss scale / learning rate after every other workgroup has read the old ones.  work_dev[BEZ_PPO_ADAM_WORK_FLOATS]:
 * [0] is that ticket counter -- ZERO on entry (allocate it zeroed), zero again on return (no memset per step).  grads_dev must be
 * 16-byte aligned.
 * params_f16_dev (NULL or n fp16): receives the updated parameters as fp16 in the same pass (the AMP working copy).
 * ntail (0..4) bookkeeping sums ride in the last launch: *tail_dst_dev[i] += *tail_src_dev[i] * tail_scale[i] (host arrays of
 * device pointers / host floats) -- the epoch's KL and loss accumulators of a2c_common.py's train_epoch [ext].
 * adapt_kl_dev (NULL = off): after the step, *lr_dev moves by rl_games' AdaptiveScheduler rule on *adapt_kl_dev (lr /= 1.5 above
 * 2 x adapt_kl_threshold, floor min_lr; lr *= 1.5 below half of it, cap max_lr) -- the 'legacy' schedule, once per minibatch step.
 * extra (NULL = none): work of neighbouring launches folded into this one -- (a) the fragment-major fp16 weight copies the MFMA
 * policy kernels read: packed_f16_dev[map_a_dev[i]] = packed_f16_dev[map_b_dev[i]] = fp16(param i) (negative map entry: no copy), what
 * bez_ppo_scatter2_f16 does from params_f16_dev; (b) the input normaliser's update for the NEXT minibatch, what bez_ppo_rms_apply
 * does from rms_moments_dev (rms_cols <= 1024); (c) the squared norm and non-finite count taken from the per-workgroup shares the
 * gradient's producer left instead of reading the gradient once more in every workgroup. */
typedef struct BezPpoAdamExtra {
  const float* norm_parts_dev; int32_t norm_parts; /* (c) the norm_parts pairs bez_ppo_grad_reduce_all (or bez_ppo_grad_norm_parts) left for THIS gradient */
  float grad_div;                                  /* data parallel: the gradient buffer holds the SUM over this many ranks (an all-reduce); 0 = 1.
                                                      Folded into the unscale factor: no separate division pass over the buffer */
  float* grid_norm_dev;                            /* data parallel, instead of norm_parts_dev: BEZ_PPO_ADAM_GRIDNORM_FLOATS zero-initialised floats; the launch
                                                      forms the norm itself -- every workgroup sums its own slice, the workgroups meet at a counter in this
                                                      buffer (all of them are co-resident: <= 256 workgroups), each adds the shares in the same fixed order --
                                                      and leaves the counter at zero.  No extra launch, no workgroup reads the whole gradient */
  const int32_t* map_a_dev; const int32_t* map_b_dev; void* packed_f16_dev;
  const double* rms_moments_dev; int32_t rms_cols; double* rms_mean_dev; double* rms_var_dev; double* rms_count_dev;
} BezPpoAdamExtra;
/* Data parallel: an all-reduce has replaced the gradient bez_ppo_grad_reduce_all left its norm shares for.  One small launch re-forms them
 * (workgroup b: sum g^2 and non-finite count of its slice of the n-element buffer, fixed order) so that the optimiser launch reads `parts`
 * pairs instead of the whole gradient in every workgroup.  parts_dev: 2 * parts floats; returns the number of pairs written (<= parts). */
int bez_ppo_grad_norm_parts(const float* grads_dev, int64_t n, float* parts_dev, int32_t parts, void* stream);
#define BEZ_PPO_ADAM_WORK_FLOATS 258
#define BEZ_PPO_ADAM_GRIDNORM_FLOATS 516
/* How many workgroups of bez_ppo_adam_step's launch can be resident on the current device at once (occupancy x compute units), and how
 * many the launch uses for n parameters.  With BezPpoAdamExtra.grid_norm_dev the launch's workgroups meet at a counter: that is only safe
 * when all of them are co-resident (a CU-masked or partitioned device, a smaller part, two ranks on one GPU can break the assumption) --
 * bez_ppo_adam_step then returns -6 instead of launching, and the caller falls back to bez_ppo_grad_norm_parts (PPO ABI 9). */
int bez_ppo_adam_grid_capacity(int64_t n, int32_t* resident_workgroups, int32_t* launch_workgroups);
int bez_ppo_adam_step(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t n, float* steps_dev,
                      int32_t nsteps, float* lr_dev, float beta1, float beta2, float eps, float weight_decay, float max_norm, float* scale_dev,
                      int32_t* growth_tracker_dev, float growth_factor, float backoff_factor, int32_t growth_interval, float* work_dev,
                      void* params_f16_dev, int32_t ntail, float* const* tail_dst_dev, const float* const* tail_src_dev, const float* tail_scale,
                      const float* adapt_kl_dev, float adapt_kl_threshold, float min_lr, float max_lr, const BezPpoAdamExtra* extra, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BEZ_SIM_H */
