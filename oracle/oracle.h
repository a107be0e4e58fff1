/* oracle.h -- CPU restatement of the orcdchomp CHOMP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * import, link or execute it.  The product path (or_cdchomp_amd/) never does.
 *
 * What this restates (reference = /root/reference, personalrobotics/or_cdchomp):
 *   grid.c / grid_flood.c   -> ora_grid_*        src/libcd/grid.c, grid_flood.c
 *   kin.c / spatial.c       -> ora_kin_*, ora_spatial_*   src/libcd/kin.c, spatial.c
 *   chomp.c                 -> ora_chomp_*       src/libcd/chomp.c
 *   sphere_cost_pre/_cost   -> ora_sphere_cost*  src/orcdchomp_mod.cpp:968-1327
 *   create/iterate/gettraj  -> ora_run_*         src/orcdchomp_mod.cpp:1800-3011
 *   GSL mt19937 + gaussian  -> ora_rng_*         (third party, absent; published algorithm)
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   PINNED by oracle/_ref (reference sources compiled in this container):
 *     grid lookup/interp/grad/sedt/bin_sdf, flood fill, shparse tokenizer.
 *   PINNED by known answers recorded from the reference build in SURVEY.md 8(c):
 *     A, Ainv, B, trC, Kvels entries; grid probe values.
 *   PARITY UNPINNED (reference source needs cblas/lapacke/OpenRAVE/GSL, none of
 *   which exist in this image, so the reference cannot be run):
 *     chomp.c iterate as a whole, kin.c/spatial.c, sphere_cost*, FK/Jacobians
 *     (OpenRAVE), the GSL noise stream.  These follow the reference text line by
 *     line with the file:line cited at every function.  They are checked against a
 *     SECOND restatement written independently in numpy from the same text (dense
 *     matrices and LAPACK for the optimizer and the constraint step, finite
 *     differences of a numpy kinematics for every Jacobian, numpy's mt19937 for
 *     the noise): tests/test_oracle_random_robots.py.  That is two readings of
 *     the reference agreeing, not the reference run: "parity unpinned" stands.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ grid */
/* src/libcd/grid.h:29-41 (3-d, double cells only: the only use on the path) */
typedef struct ora_grid
{
   int n;            /* always 3 */
   int sizes[3];
   size_t ncells;
   double lengths[3];
   double * data;    /* C order: data[(x*NY+y)*NZ+z]  (grid.c:31-32) */
} ora_grid;

ora_grid * ora_grid_create(const int sizes[3], const double lengths[3], double init);
ora_grid * ora_grid_copy(const ora_grid * g);
void ora_grid_free(ora_grid * g);
int ora_grid_index_to_subs(const ora_grid * g, size_t index, int * subs);
int ora_grid_center_index(const ora_grid * g, size_t index, double * center);
int ora_grid_lookup_index(const ora_grid * g, const double * p, size_t * index);
int ora_grid_double_grad(const ora_grid * g, const double * p, double * grad);
int ora_grid_double_interp(const ora_grid * g, const double * p, double * valuep);
int ora_grid_double_dt_sqeuc(ora_grid ** gp_dt, const ora_grid * g_func);
int ora_grid_double_bin_sdf(ora_grid ** gp_dt, const ora_grid * g_emp);
/* flood fill specialised to the only callback the module uses
 * (replace_1_to_0, src/orcdchomp_mod.cpp:160-168); returns #cells replaced */
long ora_grid_flood_fill_1_to_0(ora_grid * g, size_t index_start);

/* ------------------------------------------------------------------- kin */
int ora_kin_pose_identity(double pose[7]);
int ora_kin_pose_normalize(double pose[7]);
int ora_kin_pose_compose(const double pose_ab[7], const double pose_bc[7], double pose_ac[7]);
int ora_kin_pose_compos(const double pose_ab[7], const double pos_bc[3], double pos_ac[3]);
int ora_kin_pose_compose_vec(const double pose_ab[7], const double vec_bc[3], double vec_ac[3]);
int ora_kin_pose_invert(const double pose_in[7], double pose_out[7]);
int ora_kin_quat_to_R(const double quat[4], double R[3][3]);
int ora_spatial_xm_from_pose(double xm[6][6], const double pose[7]);
int ora_spatial_pose_jac(const double pose[7], double jac[6][7]);
/* the pieces the TSR constraint needs (src/libcd/kin.c:418-459,510-517,615-646,682-717; spatial.c:339-375) */
int ora_kin_quat_from_R(double quat[4], double R[3][3]);
int ora_kin_pose_from_dR(double pose[7], const double d[3], double R[3][3]);
int ora_kin_pose_to_xyzypr(const double pose[7], double xyzypr[6]);
int ora_kin_pose_to_xyzypr_J(const double pose[7], double J[6][7]);
int ora_spatial_pose_jac_inverse(const double pose[7], double jac_inverse[7][6]);

/* ------------------------------------------------------------------- rng */
/* GSL gsl_rng_mt19937 / gsl_rng_uniform / gsl_ran_gaussian restated */
typedef struct ora_rng { unsigned long mt[624]; int mti; } ora_rng;
void ora_rng_set(ora_rng * r, unsigned long seed);
unsigned long ora_rng_get(ora_rng * r);
double ora_rng_uniform(ora_rng * r);
double ora_ran_gaussian(ora_rng * r, double sigma);

/* ----------------------------------------------------------------- chomp */
/* a per-point hard constraint (struct cd_chomp_con, src/libcd/chomp.h:27-36) */
struct ora_chomp;
typedef struct ora_chomp_con
{
   struct ora_chomp_con * next;
   int k, i;
   void * cptr;
   int (*con_eval)(void * cptr, struct ora_chomp * c, int i, double * point, double * con_val, double * con_jacobian);
   double * h, * J;
} ora_chomp_con;

/* src/libcd/chomp.h:38-101 */
typedef struct ora_chomp
{
   int m, n;
   double * T; int ldt;
   double ** T_points;
   double * G; double ** G_points;
   double * AG; double ** AG_points;
   int D;
   double * wds;
   double * initsfinals;
   double ** inits;
   double ** finals;
   double dt;
   double * A, * Ainv, * B;
   double trC;
   double * jlimit_lower, * jlimit_upper;
   double * Gjlimit, * GjlimitAinv;
   double * cost_nxn, * cost_mxn;
   double * Kvels, * Evels, * vels;
   void * cptr;
   int (*cost_pre)(void * cptr, struct ora_chomp * c, int m, double ** T_points);
   int (*cost)(void * cptr, struct ora_chomp * c, int ti, double * point, double * vel, double * costp, double * grad);
   double lambda;
   int use_momentum;
   int leapfrog_first;
   int last_num_limadjs;   /* instrumentation only: rounds of the joint-limit loop */
   /* hard constraints (chomp.h:83-90); the list is LIFO like the reference's */
   ora_chomp_con * cons;
   int cons_k;
   double * cons_h, * cons_Jcol, * cons_JAJT, * cons_delta;
   int * cons_ipiv;
   int cons_error;         /* instrumentation only: LU met a zero pivot ("constraint inversion error!") */
} ora_chomp;

int ora_chomp_create(ora_chomp ** cp, int m, int n, int D, double * T, int ldt);
void ora_chomp_free(ora_chomp * c);
int ora_chomp_init(ora_chomp * c);
/* src/libcd/chomp.c:219-234; ora_chomp_alloc_constraints is the tail of cd_chomp_init (chomp.c:405-425),
 * callable again after constraints were added to an initialised solver */
int ora_chomp_add_constraint(ora_chomp * c, int k, int i, void * cptr,
   int (*con_eval)(void * cptr, struct ora_chomp * c, int i, double * point, double * con_val, double * con_jacobian));
int ora_chomp_alloc_constraints(ora_chomp * c);
int ora_chomp_iterate(ora_chomp * c, int do_iteration, double * costp_total, double * costp_obs, double * costp_smooth);

/* ----------------------------------------------------------- robot model */
/* The build's own kinematic model (OpenRAVE stand-in).
 * Links are in topological order (parent index < own index, root parent -1).
 * link frame = parent link frame o pose_parent_joint o motion(axis, q[dof]).
 * joint_type: 0 fixed, 1 revolute, 2 prismatic. */
struct ora_grabbed;
typedef struct ora_robot
{
   int n_links;
   const int * parent;            /* [n_links] */
   const double * pose_parent_joint; /* [n_links][7] x y z qx qy qz qw */
   const int * joint_type;        /* [n_links] */
   const double * axis;           /* [n_links][3], joint frame */
   const int * dof_index;         /* [n_links], robot dof or -1 */
   int n_dof;
   const double * limit_lower;    /* [n_dof] */
   const double * limit_upper;    /* [n_dof] */
   /* spheres in <orcdchomp><spheres> XML order (src/orcdchomp_kdata.cpp:79-94) */
   int n_spheres;
   const int * sphere_link;       /* [n_spheres] */
   const double * sphere_pos;     /* [n_spheres][3] in link frame */
   const double * sphere_radius;  /* [n_spheres] */
   int n_adjacent;                /* link pairs the robot description declares adjacent (<adjacent> tags) */
   const int * adjacent;          /* [n_adjacent][2] */
   /* kinbodies the robot is grabbing, in GetGrabbed() order (src/orcdchomp_mod.cpp:2168-2171); 0 / NULL: none */
   int n_grabbed;
   const struct ora_grabbed * grabbed;
} ora_robot;

/* A grabbed kinbody as create sees it (src/orcdchomp_mod.cpp:2173-2210): its <orcdchomp> spheres, the robot link
 * that holds it (RobotBase::IsGrabbing(k)) and the world transform of the kinbody's link at create
 * (k->GetLink(linkname)->GetTransform(); the stand-in environment's kinbodies have one link).  A sphere rides on
 * the grabbing link at T_w_rlink^-1 * T_w_klink * pos (2200-2208). */
typedef struct ora_grabbed
{
   int robot_link;
   double pose_world_klink[7];
   int n_spheres;
   const double * sphere_pos;     /* [n_spheres][3] in the kinbody link's frame */
   const double * sphere_radius;  /* [n_spheres] */
   /* the robot's state at the moment of RobotBase::Grab (the body is rigid with its link from then on): what the body touched
    * THEN is what CheckSelfCollision leaves it out against.  has_grab_state 0: the body was grabbed in the state of create. */
   int has_grab_state;
   double grab_base_pose[7];
   const double * grab_dofvals;   /* [n_dof] */
} ora_grabbed;

/* FK: world transforms of all links.  R[n_links][9] row-major, t[n_links][3].
 * Also world joint axis / anchor per link (axis_w zero for fixed joints). */
void ora_robot_fk(const ora_robot * rob, const double base_pose[7], const double * dofvals,
   double * R, double * t, double * axis_w, double * anchor_w);
/* DoesAffect(dof, link) semantics: dof's joint is on the path root->link */
int ora_robot_does_affect(const ora_robot * rob, int dof, int link);

/* ------------------------------------------------------------------- run */
typedef struct ora_rsdf { double pose_world_gsdf[7]; double pose_gsdf_world[7]; const ora_grid * grid; } ora_rsdf;

typedef struct ora_run_params
{
   int n_points;            /* default 101 */
   int floating_base;
   double lambda;           /* default 10 */
   int D;                   /* derivative, default 1 */
   int use_momentum;
   int use_hmc;
   double hmc_resample_lambda; /* 0.02 */
   unsigned int seed;
   double epsilon, epsilon_self, obs_factor, obs_factor_self; /* 0.1 0.04 200 10 */
   /* `start_tsr`: the start point becomes a variable held on a TSR by a hard constraint
    * (src/orcdchomp_mod.cpp:1988-1992, 2316-2323, 2482-2499, 2570-2576); the fields as for ora_run_add_contsr */
   int start_tsr;           /* default 0 */
   int start_ee_link;
   double start_tool[7], start_T0w[7], start_Twe[7], start_Bw[12];
} ora_run_params;
void ora_run_params_default(ora_run_params * p);

typedef struct ora_run ora_run;

/* create: src/orcdchomp_mod.cpp:1800-2688.  robot state = base_pose + dofvals
 * (all dofs) + active dof indices.  adofgoal[n_adof]; basegoal[7] or NULL.
 * sdfs: per-field grid + pose of the grid in the world (already composed with
 * the kinbody transform, mod.cpp:2359-2367).  Returns NULL + message on error. */
ora_run * ora_run_create(const ora_robot * rob, const double base_pose[7], const double * dofvals,
   int n_adof, const int * adofindices, const double * adofgoal, const double * basegoal,
   int n_sdfs, const ora_grid * const * grids, const double * poses_world_gsdf /* [n_sdfs][7] */,
   const ora_run_params * params, const char ** errmsg);
/* iterate: src/orcdchomp_mod.cpp:2690-2852.  costs_out[3] = total, obs, smooth of the
 * FINAL evaluation (do_iteration=0); trace (optional) [n_iter][3] per-iteration costs.
 * returns 0, or -1 if the joint-limit loop ran out ("Resulting trajectory is outside of joint limits!") */
/* TSR hard constraint on every moving point (`con_tsr all ...` / `everyn_tsr`, src/orcdchomp_mod.cpp:1330-1657,
 * 2466-2480,2582-2612).  ee_link + tool[7] (pose of the end effector in that link: identity for `link NAME`,
 * the manipulator's local tool transform for `manipee`), T0w[7], Twe[7] and Bw[6][2] of the TSR
 * (tsr_create_parse, mod.cpp:3068-3111).  Call in the reference's order: everyn_tsr first, then the con_tsrs. */
int ora_run_add_contsr(ora_run * r, int ee_link, const double tool[7], const double T0w[7], const double Twe[7], const double * Bw);
/* the constraint value and Jacobian of constraint `which` (order of addition) at a trajectory row: h[k], J[k][n] */
int ora_run_eval_contsr(ora_run * r, int which, const double * point, double * h, double * J);
int ora_run_iterate(ora_run * r, int n_iter, double * costs_out, double * trace);
/* as above but with externally supplied momentum noise: noise[k][m][n] is used for the
 * k-th resample of this call instead of the run's own rng stream */
int ora_run_iterate_noise(ora_run * r, int n_iter, double * costs_out, double * trace,
   const double * noise, int n_noise);
void ora_run_destroy(ora_run * r);
/* accessors */
int ora_run_n(const ora_run * r);
int ora_run_m(const ora_run * r);
int ora_run_n_points(const ora_run * r);
int ora_run_n_spheres_active(const ora_run * r);
int ora_run_n_spheres(const ora_run * r);
double * ora_run_traj(ora_run * r);          /* [n_points][n] */
ora_chomp * ora_run_chomp(ora_run * r);
int ora_run_hmc_resample_iter(const ora_run * r);
int ora_run_iter(const ora_run * r);          /* r->iter: where the last iterate call stopped */
void ora_run_set_traj(ora_run * r, const double * traj);   /* overwrite [n_points][n] (tests) */
/* the collision re-check of mod::gettraj (src/orcdchomp_mod.cpp:2958-3006), see ora_run.c */
int ora_run_collision_recheck(ora_run * r, const double * vmax, int * collides, double * time_out,
   int * sphere_out, int * field_out, double * depth_out);
/* the starttraj sampling of mod::create (src/orcdchomp_mod.cpp:2375-2416), see ora_run.c */
void ora_sample_starttraj(int count, int dof, const double * wp, const double * deltatime, int n_points, double * out);
/* ... with floating_base (src/orcdchomp_mod.cpp:2378-2404) and gettraj's affine_transform / affine_velocities groups
 * (src/orcdchomp_mod.cpp:2912-2949), see ora_run.c */
void ora_sample_starttraj_floating(int count, int n_adof, const double * wp_joint, const double * wp_base, const double * deltatime,
   int n_points, double * out);
void ora_gettraj_affine_groups(const double * traj, int n_points, int n, const double * deltatime, double * out);
/* link pairs a self-collision check skips (the sphere model's stand-in for OpenRAVE's adjacent links), see ora_run.c */
void ora_robot_self_pairs(const ora_robot * rob, unsigned char * excl);
/* one evaluation of sphere_cost_pre + sphere_cost for every moving point on the
 * current trajectory: G[m][n] (unscaled, as the callback leaves it), costs[m],
 * sphere_poss_all[n_points][S_a][3] */
int ora_run_eval_obstacle(ora_run * r, double * G, double * costs, double * sphere_poss_all);
/* active-first sphere order: fills idx[n_spheres] with XML indices */
void ora_run_sphere_order(const ora_run * r, int * idx);
/* [n][n] (n = the run's spheres, XML order through the robot and the held bodies): 1 = the re-check's self-collision leg never
 * tests the pair (taken at create: links that are adjacent or touch at the zero pose for the robot's own spheres; a held body
 * against its holder link and against what it touched in the configuration of create) */
void ora_run_self_excluded(const ora_run * r, unsigned char * excl);

/* batch driver for the CPU baseline: runs create+iterate for n_runs goals
 * (OpenMP over runs when built with -fopenmp).  traj_out [n_runs][n_points][n],
 * costs_out [n_runs][3], status_out [n_runs].  Returns threads used. */
int ora_batch_run(const ora_robot * rob, const double base_pose[7], const double * dofvals,
   int n_adof, const int * adofindices, int n_runs, const double * adofgoals, const double * basegoals,
   int n_sdfs, const ora_grid * const * grids, const double * poses_world_gsdf,
   const ora_run_params * params, const unsigned int * seeds, int n_iter,
   double * traj_out, double * costs_out, int * status_out, int max_threads);

/* LAPACKE_dgesv(LAPACK_ROW_MAJOR, n, 1, A, n, ipiv, b, 1) as the constraint step uses it (src/libcd/chomp.c:579-581):
 * returns info; info > 0 (singular): b is left as it was */
int ora_dgesv_one(int n, double * A, int * ipiv, double * b);

/* --------------------------------------------------------------- shparse */
/* src/libcd/util_shparse.c:37-128 */
int ora_util_shparse(char * in, int * argcp, char *** argvp);

#ifdef __cplusplus
}
#endif
#endif
