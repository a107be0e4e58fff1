/* orcdchomp_amd.h -- C ABI of the MI355X-native CHOMP hot path.
 *
 * Drop-in boundary for the path  computedistancefield -> create -> iterate ->
 * gettraj -> destroy  of the orcdchomp OpenRAVE module (personalrobotics/or_cdchomp).
 * Plain C: pointers and sizes only, no C++/torch types.  Every entry point names
 * the reference interface it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - all functions return 0 on success, non-zero on error; the message of the
 *     last error of a module is available through orc_last_error() and uses the
 *     reference's own exception strings.  1 = the call failed
 *     (message set), 2 = no module was passed.  A null pointer where an array, a
 *     name or a struct is required, an unknown handle, a short buffer or a
 *     malformed robot description is such an error, not a fault; pointers
 *     documented as optional may be NULL.
 *   - pose = 7 doubles [x y z qx qy qz qw]           (src/libcd/kin.c:42-52)
 *   - grids are C ordered [x][y][z] doubles            (src/libcd/grid.c:31-32)
 *   - trajectories are run-major: traj[run][waypoint][dof]
 *   - handles returned as text by create are opaque strings, as in the reference
 *     ("%p" there, an integer id here; src/orcdchomp_mod.cpp:2670-2674).
 */
#ifndef ORCDCHOMP_AMD_H
#define ORCDCHOMP_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_module orc_module;

/* ---- module lifetime ---------------------------------------------------------
 * replaces the plugin entry points CreateInterfaceValidated / DestroyPlugin
 * (src/orcdchomp.cpp:50-74) and the mod constructor/destructor
 * (src/orcdchomp_mod.h:45-75).  device = HIP device ordinal (one process per GPU). */
orc_module * orc_module_new(int device);
/* one module over several GPUs of the node, inside one process: every batch is cut into contiguous
 * blocks of runs, one per entry of `devices` (an ordinal may repeat: its blocks then run on streams
 * of their own), each block iterates on its device and copies its results straight into the
 * caller's arrays -- the host-side gather; no collective.  Fields and the robot model
 * are replicated to every device at create.  orc_module_new(d) is the list {d}. */
orc_module * orc_module_new_multi(const int * devices, int n_devices);
void orc_module_free(orc_module * mod);
const char * orc_last_error(const orc_module * mod);
/* HIP stream all kernels and copies of this module are issued on (a hipStream_t,
 * e.g. torch.cuda.current_stream().cuda_stream); NULL = the default stream. */
int orc_set_stream(orc_module * mod, void * hip_stream);
/* give the module a pool of n internal streams per device: batches created afterwards are bound to
 * them round-robin, so launches of independent batches overlap on the GPU (0 = back to one stream).
 * Fails while batches exist: they hold the streams they were bound to. */
int orc_set_num_streams(orc_module * mod, int n);

/* ---- SendCommand -------------------------------------------------------------
 * replaces ModuleBase::SendCommand -> orcwrap_call (src/orcwrap.cpp:37-69) ->
 * mod::computedistancefield / addfield_fromobsarray / removefield / create /
 * iterate / gettraj / destroy (src/orcdchomp_mod.h:58-66).  Same command names,
 * same argv grammar (shell-style quoting, src/libcd/util_shparse.c), same textual
 * returns.  out receives the reply (NUL terminated, truncated to out_cap).
 * Batch extensions (additive): createbatch, iteratebatch, gettrajbatch.
 * create ... dat_filename PATH writes the reference's per-iteration log "%d %f %f %f %f\n"
 * (iteration, seconds, cost_total, cost_obs, cost_smooth; mod.cpp:2306-2310, 2811-2818); the seconds
 * are wall seconds since the iterate call began (the reference: thread CPU seconds), a fused launch's
 * interval divided evenly over its iterations.  createbatch takes a pattern with %d = run index, and
 * `devices 'i j ...'` to shard one batch over GPUs (see orc_module_new_multi). */
int orc_send_command(orc_module * mod, const char * cmd, char * out, size_t out_cap);
/* size in bytes (without NUL) of the full reply of the last successful command */
size_t orc_last_reply_size(const orc_module * mod);
/* copy of the full reply of the last successful command (for replies larger than out_cap) */
int orc_last_reply(const orc_module * mod, char * out, size_t out_cap);

/* ---- environment stand-ins ---------------------------------------------------
 * The reference reads robots and bodies from the OpenRAVE environment
 * (e->GetRobot / e->GetKinBody, src/orcdchomp_mod.cpp:1894,336).  OpenRAVE is
 * third party; these calls hand the same information over explicitly. */

/* kinematic tree, links in topological order (parent index < own index):
 *   link frame = parent link frame o pose_parent_joint o motion(axis, q[dof])
 * joint_type 0 fixed, 1 revolute, 2 prismatic.  Spheres in <orcdchomp><spheres>
 * XML order (src/orcdchomp_kdata.cpp:79-94, struct sphere src/orcdchomp_kdata.h:33-39). */
typedef struct orc_robot_desc
{
   int n_links;
   const int * parent;               /* [n_links], -1 for the root */
   const double * pose_parent_joint; /* [n_links][7] */
   const int * joint_type;           /* [n_links] */
   const double * axis;              /* [n_links][3] unit, in the joint frame */
   const int * dof_index;            /* [n_links] robot dof, -1 for fixed */
   int n_dof;
   const double * limit_lower;       /* [n_dof]  (GetDOFLimits, mod.cpp:2639) */
   const double * limit_upper;       /* [n_dof] */
   int n_spheres;
   const int * sphere_link;          /* [n_spheres] link index */
   const double * sphere_pos;        /* [n_spheres][3] in the link frame */
   const double * sphere_radius;     /* [n_spheres] */
} orc_robot_desc;

int orc_env_add_robot(orc_module * mod, const char * name, const orc_robot_desc * desc);
int orc_robot_set_transform(orc_module * mod, const char * name, const double pose[7]);      /* robot->SetTransform */
int orc_robot_set_dof_values(orc_module * mod, const char * name, const double * values, int n); /* SetDOFValues */
int orc_robot_set_active_dofs(orc_module * mod, const char * name, const int * indices, int n); /* SetActiveDOFs */
/* GetDOFVelocityLimits: used by the linear retimer of gettraj (default 1 for every dof) */
int orc_robot_set_velocity_limits(orc_module * mod, const char * name, const double * limits, int n);

/* Workgroup shape of the iterate kernel for the batches created from now on: 0 = the planner's choice
 * (256 threads, three workgroups per CU for the WAM), 192 = three wavefronts, four workgroups per CU:
 * 1024 runs are then resident at once on 256 CUs and ONE launch of 1024 runs ends ~10 % earlier; large
 * or overlapping batches are ~5 % slower with it; 512 = eight wavefronts on one run, one run per CU, the
 * whole trajectory in one tile: the latency shape for batches smaller than the chip (one run: 34 k
 * instead of 22 k iterations/s, 256 runs: 5.4 M instead of 4.2 M).  The single-run `create` command uses
 * 512 unless a shape is set here.  128 = two wavefronts on a run, up to eight runs per CU (fp64 fixed-base chains with at
 * most 16 active spheres): what the planner gives runs with TSR constraints -- whose elimination is the work of two
 * wavefronts -- when the module's launches overlap (orc_set_num_streams >= 2): +18 %.  The shape never depends on the batch
 * itself, so that a run's result does not depend on what shares its batch (trajectories are bit-identical across shapes;
 * robots with 17 .. 32 active spheres have the 256- and 512-thread shapes and fall back to the many-sphere kernels at 192). */
int orc_set_workgroup_threads(orc_module * mod, int threads);
/* Register budget of the batches created from now on: 0 (default) the planner's choice, 3 the kernels' own (three 256-thread
 * workgroups per CU at 168 registers for fp64), 4: four per CU at 128 registers with smaller tiles, where a kernel is built
 * for it (fp64 robots on a fixed-base chain with at most 16 active spheres -- the WAM of the BASELINE configurations -- or
 * with 17 .. 32: the robot that holds something; others, and runs too long for the smaller share of the LDS, keep their
 * default).  The planner's choice is a function of the robot, the run parameters and this module's settings, never of the
 * batch: four per CU for runs with TSR constraints and for the 17 .. 32-sphere family (faster whatever the launch pattern)
 * and, when the module's launches overlap (orc_set_num_streams >= 2), for every fixed-base chain (+3-5 %; one launch of
 * <= 1024 runs at a time is 3 % faster at three).  Trajectories are bit-identical either way. */
int orc_set_workgroups_per_cu(orc_module * mod, int workgroups);

/* What the TSR constraints of `create` address on the robot (src/orcdchomp_mod.cpp:1957-1976):
 * GetLink(name) for `con_tsr 'all link NAME'`; GetManipulators() / GetActiveManipulator() and their
 * GetEndEffectorTransform() (= end-effector link transform o tool_pose) for `'all manipee NAME'`,
 * `'all'` and `everyn_tsr`.  The first manipulator added is the active one. */
int orc_robot_set_link_names(orc_module * mod, const char * name, const char * const * names, int n);
int orc_robot_add_manipulator(orc_module * mod, const char * name, const char * manip, int ee_link, const double tool_pose[7]);
int orc_robot_set_active_manipulator(orc_module * mod, const char * name, const char * manip);
/* Pairs of links the robot description declares adjacent (OpenRAVE robot files: <adjacent>linkA linkB</adjacent>;
 * KinBody::GetAdjacentLinks): RobotBase::CheckSelfCollision, which the re-check of mod::gettraj calls
 * (src/orcdchomp_mod.cpp:2998-2999), never tests them.  link_pairs [n_pairs][2] link indices. */
int orc_robot_set_adjacent_links(orc_module * mod, const char * name, const int * link_pairs, int n_pairs);
/* The self-collision leg of the re-check is a stand-in: OpenRAVE tests the links' meshes, this library the bounding
 * spheres of the optimizer's model, which are fatter (a trajectory whose meshes clear each other by less than the spheres'
 * slack is "in collision" here and returned by the reference), and "adjacent in the initial configuration" is taken with
 * all dofs at zero.  enabled = 0 leaves that leg out for this robot (gettraj, gettrajbatch ... verdict,
 * orc_batch_collision_verdict); `gettraj ... no_self_collision_check` does the same for one call.  Default 1. */
int orc_robot_set_self_check(orc_module * mod, const char * name, int enabled);

/* a kinbody made of oriented boxes (InitFromBoxes-style); box_poses [n_boxes][7]
 * in the kinbody frame, half_extents [n_boxes][3] */
int orc_env_add_kinbody_boxes(orc_module * mod, const char * name, int n_boxes,
   const double * box_poses, const double * half_extents);
/* a kinbody given as a triangle mesh (KinBody::InitFromTrimesh; the reference's own scene is meshes, scripts/test_wam7.py:23-28):
 * vertices [n_tri][3][3] in the kinbody frame.  computedistancefield sweeps its cube against the triangles (the collision
 * query of src/orcdchomp_mod.cpp:462-531; a mesh is a surface, the flood fill of 540-548 closes its inside; touching counts as
 * a collision so that a closed mesh gives a closed shell of cells).  Called for an existing kinbody of boxes the triangles are
 * added to it. */
int orc_env_add_kinbody_trimesh(orc_module * mod, const char * name, int n_tri, const double * vertices);
/* KinBody::SetTransform.  A kinbody the robot holds is moved there and rides with its link from there on (the grab's
 * relative transform is taken anew; passing the pose orc_body_get_transform returns changes nothing) */
int orc_kinbody_set_transform(orc_module * mod, const char * name, const double pose[7]);
int orc_kinbody_enable(orc_module * mod, const char * name, int enabled);
/* KinBody::GetTransform of a robot or kinbody; a kinbody the robot holds is where its link carries it now */
int orc_body_get_transform(orc_module * mod, const char * name, double pose_out[7]);

/* ---- grabbed bodies ----------------------------------------------------------
 * mod::create collects the spheres of the robot AND of every kinbody the robot is grabbing
 * (src/orcdchomp_mod.cpp:2168-2300: r->robot->GetGrabbed(), the body's <orcdchomp> kdata, the link
 * r->robot->IsGrabbing(k)); carrying an object is the planner's normal use.  The stand-ins for what the
 * reference asks OpenRAVE: */
/* the <orcdchomp><spheres> data of a kinbody (src/orcdchomp_kdata.cpp:79-94; the stand-in's kinbodies have one
 * link, so the sphere's link attribute is that link): sphere_pos [n_spheres][3] in the kinbody frame */
int orc_kinbody_set_spheres(orc_module * mod, const char * name, int n_spheres, const double * sphere_pos,
   const double * sphere_radius);
/* RobotBase::Grab(body, link): from now on the kinbody is rigid with robot link `link` at its current relative
 * transform (orc_robot_set_dof_values / orc_robot_set_transform carry it along) and a `create` for this robot
 * adds the body's spheres to the run: each rides on `link` at T_w_rlink^-1 o T_w_klink o pos
 * (src/orcdchomp_mod.cpp:2200-2208), active when an active dof moves `link` (2265-2291); the run's sphere list is the
 * last grabbed body's spheres first and the robot's last, as the head insertion of 2273-2290 leaves it.  XML
 * sphere indices reported by the collision verdict count through the robot's spheres, then the held bodies' in the
 * order they were grabbed.  A held body (or the robot) without spheres makes `create` fail with the reference's
 * "no spheres! kinbody does not have a <orcdchomp> tag defined?" (2262-2263).  The re-check of gettraj includes the
 * spheres a run was created with (the reference's note at 2992-2996).  A run keeps the spheres it was created with. */
int orc_robot_grab(orc_module * mod, const char * robot, const char * kinbody, int link);
/* RobotBase::Release(body) / ReleaseAllGrabbed(): the body stays where the link left it */
int orc_robot_release(orc_module * mod, const char * robot, const char * kinbody);
int orc_robot_release_all(orc_module * mod, const char * robot);

/* ---- SDF access --------------------------------------------------------------
 * the module's field list (struct sdf, src/orcdchomp_mod.cpp:148-153) */
int orc_scene_add_sdf(orc_module * mod, const char * kinbody, const int sizes[3], const double lengths[3],
   const double pose_kinbody_gsdf[7], const double * sdf_data);
/* copy out a computed field: sizes[3], lengths[3], pose[7] (grid wrt kinbody), data (may be NULL to query sizes) */
int orc_scene_get_sdf(orc_module * mod, const char * kinbody, int sizes[3], double lengths[3], double pose[7],
   double * data, size_t data_cap);

/* ---- kernel-level batch API --------------------------------------------------
 * what the command layer calls; one batch = n_runs independent CHOMP runs that
 * share robot, fields and parameters (struct run, src/orcdchomp_mod.cpp:887-966;
 * cd_chomp, src/libcd/chomp.h:38-101).  A single `create` is a batch of 1. */
typedef struct orc_batch_params
{
   int n_points;               /* default 101        (mod.cpp:1840) */
   int floating_base;          /*                    (mod.cpp:1843,1928) */
   double lambda;              /* default 10         (mod.cpp:1824) */
   int derivative;             /* D, default 1       (mod.cpp:1826) */
   int use_momentum;           /*                    (mod.cpp:1825) */
   int use_hmc;                /*                    (mod.cpp:1873) */
   double hmc_resample_lambda; /* default 0.02       (mod.cpp:1875) */
   double epsilon;             /* default 0.1        (mod.cpp:1845) */
   double epsilon_self;        /* default 0.04       (mod.cpp:1846) */
   double obs_factor;          /* default 200        (mod.cpp:1847) */
   double obs_factor_self;     /* default 10         (mod.cpp:1848) */
   int precision;              /* 64 (default) or 32: arithmetic type of the device path */
} orc_batch_params;
void orc_batch_params_default(orc_batch_params * p);

/* replaces mod::create (src/orcdchomp_mod.cpp:1800-2688) for n_runs runs at once.
 * starts: [n_runs][n_adof] or NULL (= the robot's current active dof values, as the
 * reference does); goals: [n_runs][n_adof]; basegoals: [n_runs][7] or NULL;
 * seeds: [n_runs] or NULL (0).  *batch_id receives the handle. */
int orc_batch_create(orc_module * mod, const char * robot, const orc_batch_params * params, int n_runs,
   const double * starts, const double * goals, const double * basegoals, const unsigned int * seeds,
   int * batch_id);
/* replaces mod::iterate (src/orcdchomp_mod.cpp:2690-2852): n_iter iterations of
 * cd_chomp_iterate (src/libcd/chomp.c:430-683) for every run, then the final
 * cost evaluation.  costs_out [n_runs][3] = total, obs, smooth (may be NULL);
 * status_out [n_runs]: 0 ok, -1 "Resulting trajectory is outside of joint limits!". */
int orc_batch_iterate(orc_module * mod, int batch_id, int n_iter, double * costs_out, int * status_out);
/* A run that leaves its joint limits stops iterating for the rest of THAT call (the reference throws
 * out of mod::iterate, src/orcdchomp_mod.cpp:2799-2803) and reports status -1 and the costs of its
 * last complete iteration; the run stays usable and a later call iterates it again, as in the
 * reference.  Iterations each run completed in the last call: iters_out [n_runs]. */
int orc_batch_iterations_done(orc_module * mod, int batch_id, int * iters_out);
/* asynchronous form for measurement: enqueue only, results stay on the device */
int orc_batch_iterate_async(orc_module * mod, int batch_id, int n_iter);
int orc_batch_sync(orc_module * mod, int batch_id, double * costs_out, int * status_out);
/* per-iteration cost trace of the last iterate call: [n_runs][n_iter][3] (total, obs, smooth as the
 * reference logs them, mod.cpp:2798); rows of iterations an aborted run did not complete are NaN */
int orc_batch_get_trace(orc_module * mod, int batch_id, double * trace_out, size_t cap_doubles);
/* momentum noise for HMC resampling supplied by the caller instead of the module's
 * own mt19937 stream: noise [n_runs][n_blocks][m][n] (used in resample order) */
int orc_batch_set_noise(orc_module * mod, int batch_id, const double * noise, int n_blocks);
/* replaces the waypoint export of mod::gettraj (src/orcdchomp_mod.cpp:2897-2903):
 * traj_out [n_runs][n_points][n] */
int orc_batch_gettraj(orc_module * mod, int batch_id, double * traj_out, size_t cap_doubles);
/* replaces the collision re-check of mod::gettraj (src/orcdchomp_mod.cpp:2958-3006) for all runs of
 * a batch at once, on the device: every run's trajectory is retimed at the dof velocity limits and
 * sampled every 0.04 rad of C-space distance like the reference's loop; a sample collides when an
 * active sphere penetrates a field (the optimizer's own model; OpenRAVE's mesh checker is third
 * party) or, the reference's `|| CheckSelfCollision` (src/orcdchomp_mod.cpp:2998-2999), when two spheres on
 * links that may collide overlap (not the same link, not parent and child, not links whose spheres already
 * overlap with all dofs at zero: the sphere model's stand-in for OpenRAVE's adjacent links).  Per run (any
 * output may be NULL): collides 0/1, time of the first contact, XML index of the sphere, index of the field
 * (a pair of spheres: -2 - the XML index of the other sphere), penetration depth in metres. */
int orc_batch_collision_verdict(orc_module * mod, int batch_id, int * collides_out, double * time_out,
                                int * sphere_out, int * field_out, double * depth_out);
/* optimizer state read-back for tests: which = "G", "AG", "T" ([n_runs][m][n]); "phase" ([n_runs][8] cycle counters with
 * ORC_PHASE_TIMERS=1); "plan" (8 numbers: kernel variant bits -- 512 = the dense pair-list family, 1 = a tree --, threads per
 * workgroup, LDS bytes per workgroup, tile, solve mode (2 closed-form scans, 3 band-inverse generators, 1 dense), workgroups
 * per CU, tiles, lanes per waypoint) */
int orc_batch_get_state(orc_module * mod, int batch_id, const char * which, double * out, size_t cap_doubles);
int orc_batch_dims(orc_module * mod, int batch_id, int * n_runs, int * n_points, int * n);
/* overwrite the trajectories of a batch (warm start; what `create starttraj` does for one run,
 * src/orcdchomp_mod.cpp:2375-2416): traj [n_runs][n_points][n] */
int orc_batch_set_traj(orc_module * mod, int batch_id, const double * traj, size_t count_doubles);
/* the collision report the last gettraj produced (the reference logs it, mod.cpp:3000) */
const char * orc_last_collision_details(const orc_module * mod);
/* replaces mod::destroy (src/orcdchomp_mod.cpp:3013-3066) */
int orc_batch_destroy(orc_module * mod, int batch_id);

/* ---- measurement -------------------------------------------------------------
 * average device time (ms) of the iterate kernel launches since the last reset,
 * measured with HIP events on the module's stream; count of launches */
int orc_kernel_time(orc_module * mod, double * total_ms, int * launches, int reset);

/* ---- host utilities (no GPU needed) -------------------------------------------
 * the host-side numerics of the path, callable on their own */
/* occupancy (0.0 free, HUGE_VAL obstacle) -> signed distance field, positive outside:
 * replaces cd_grid_double_bin_sdf (src/libcd/grid.c:637-687) */
int orc_host_bin_sdf(const int sizes[3], const double lengths[3], const double * occupancy, double * sdf_out);
/* flood fill from cell `start`, 1.0 -> 0.0, axis neighbours: replaces cd_grid_flood_fill with
 * replace_1_to_0 (src/libcd/grid_flood.c:30-111, src/orcdchomp_mod.cpp:160-168); in place */
int orc_host_flood_fill(const int sizes[3], double * cells, size_t start);
/* occupancy the way computedistancefield forms it (src/orcdchomp_mod.cpp:462-531): a cube of
 * half-extent cube_extent swept over the cell centres of a grid rooted at pose_world_gsdf against
 * oriented boxes given in the world (box_world_poses [n_boxes][7], half_extents [n_boxes][3]; the
 * stand-in for OpenRAVE's CheckCollision): occupancy_out gets HUGE_VAL where it touches, 1.0 elsewhere */
int orc_host_voxelize_boxes(const int sizes[3], const double lengths[3], const double pose_world_gsdf[7], double cube_extent,
   int n_boxes, const double * box_world_poses, const double * half_extents, double * occupancy_out);
/* ... and of a triangle mesh (world_vertices [n_tri][3][3]): HUGE_VAL where the cube touches a triangle */
int orc_host_voxelize_trimesh(const int sizes[3], const double lengths[3], const double pose_world_gsdf[7], double cube_extent,
   int n_tri, const double * world_vertices, double * occupancy_out);
/* tokenizer of the command grammar: replaces cd_util_shparse (src/libcd/util_shparse.c:37-128).
 * tokens are written NUL-separated into out; returns the token count or -1 if out is too small */
int orc_host_shparse(const char * in, char * out, size_t out_cap);
/* smoothness metric of cd_chomp_init (src/libcd/chomp.c:239-340, 393-403) in the form the
 * kernels use: dense A [m][m], endpoint couplings beta_s/beta_g [m], kappa[3] = kss ksg kgg
 * (trC), and A^-1 applied to rhs [m][ncols] by what the device uses: the cyclic-reduction tables (D=1), the
 * generators of the band inverse (2 <= D <= 4: rank-D semiseparable form, D prefix and D suffix scans
 * per column; replaces the dense dgetrf/dgetri inverse of src/libcd/chomp.c:393-403) or the dense
 * inverse (D > 4); any output may be NULL */
int orc_host_metric(int m, int derivative, double dt, double * A_out, double * beta_s_out, double * beta_g_out,
   double kappa_out[3], const double * rhs, int ncols, double * solve_out);
/* the same with the start point a variable (`start_tsr`: inits[0] == NULL, src/orcdchomp_mod.cpp:2572) */
int orc_host_metric_free_start(int m, int derivative, double dt, double * A_out, double * beta_s_out, double * beta_g_out,
   double kappa_out[3], const double * rhs, int ncols, double * solve_out);
/* rank of the semiseparable form of A^-1 the device applies for this metric (0: none -- derivative 1 has its closed
 * form, derivative > 4 the dense inverse); -1 on bad arguments */
int orc_host_metric_semisep_rank(int m, int derivative, double dt, int free_start);
/* GSL's default generator restated (src/orcdchomp_mod.cpp:2303-2304,2763,2767): n gaussians with
 * the given sigma from seed, then one uniform; out_gauss[n], out_uniform[1] */
int orc_host_gsl_stream(unsigned long seed, double sigma, int n, double * out_gauss, double * out_uniform);

#ifdef __cplusplus
}
#endif
#endif
