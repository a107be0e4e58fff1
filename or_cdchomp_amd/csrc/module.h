// module.h -- host side of the MI355X-native orcdchomp module.
//
// Mirrors class mod of the reference (src/orcdchomp_mod.h:38-90): the module owns
// the list of signed distance fields and the runs, and exposes the same commands
// through SendCommand.  OpenRAVE's environment (robots, kinbodies, transforms) is
// third party; the pieces of it the hot path reads are held here explicitly.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "dev_types.h"
#include "host_math.h"

namespace orc {

struct Robot                      // what the path reads from an OpenRAVE::RobotBase
{
   std::string name;
   int n_links = 0;
   std::vector<int> parent;
   std::vector<Pose> pose_parent_joint;
   std::vector<int> joint_type;
   std::vector<double> axis;      // [n_links][3]
   std::vector<int> dof_index;
   int n_dof = 0;
   std::vector<double> limit_lower, limit_upper;
   std::vector<double> limit_vel;   // GetDOFVelocityLimits, used by the retimer of gettraj (default 1)
   struct Sphere { int link; double pos[3]; double radius; int body = 0; };   // struct sphere, src/orcdchomp_kdata.h:33-39; body: 0 the robot's own, 1 + k a sphere of the k-th grabbed body (robot_for_run)
   std::vector<Sphere> spheres;   // XML order
   // what the TSR constraints address (`con_tsr 'all link NAME'`, `'all manipee NAME'`, src/orcdchomp_mod.cpp:1957-1976)
   std::vector<std::string> link_names;        // GetLink(name); empty: links are addressed as "link<i>"
   struct Manip { std::string name; int link; Pose tool; };   // GetEndEffectorTransform = link transform o tool
   std::vector<Manip> manips;
   int active_manip = 0;                       // GetActiveManipulator
   std::vector<std::pair<int, int>> adjacent;  // link pairs the robot description declares adjacent (<adjacent> tags)
   bool self_check = true;                     // the sphere-pair stand-in for CheckSelfCollision in gettraj's re-check (orc_robot_set_self_check)
   // kinbodies the robot holds, in the order they were grabbed (RobotBase::Grab / GetGrabbed, src/orcdchomp_mod.cpp:2168-2171):
   // the body is rigid with `link` from the moment of the grab, `rel` = T_w_link^-1 o T_w_body at that moment
   // touch_link / touch_body: what the body's spheres overlapped AT THE MOMENT OF THE GRAB (Module::grab; taken anew when
   // set_kinbody_transform re-anchors it): links of the robot, by its own spheres, and bodies the robot held already.
   // OpenRAVE's CheckSelfCollision leaves a grabbed body out against exactly those (and against the grabbing link).
   struct Grab { std::string body; int link; Xform rel; std::vector<unsigned char> touch_link; std::vector<std::string> touch_body; };
   std::vector<Grab> grabbed;
   // state
   Pose transform;
   std::vector<double> dof_values;
   std::vector<int> active_dofs;
   bool does_affect(int dof, int link) const;
   // link pairs a self-collision check skips [n_links][n_links]: the same link, parent and child, the pairs the robot
   // description declares adjacent, and links whose spheres already overlap with all dofs at zero (KinBody computes
   // its non-adjacent links from the initial configuration the same way)
   std::vector<unsigned char> self_pairs_excluded() const;
   // ... sphere by sphere for a run's list (the robot's spheres, then those of the bodies it holds, `spheres` as robot_for_run
   // leaves them, the robot in the configuration of `create`): [n][n], 1 = the pair is never tested
   std::vector<unsigned char> run_self_pairs_excluded(int n_own) const;
   // world frames of all links for the given state
   void fk(const Pose & base, const std::vector<double> & q, std::vector<Xform> & frames) const;
};

struct KinBody                    // a kinbody of oriented boxes (InitFromBoxes style) and / or triangles (a mesh: KinBody::InitFromTrimesh, the .iv files of the reference's scene)
{
   std::string name;
   Pose transform;
   bool enabled = true;
   struct B { Pose pose; double half[3]; };
   std::vector<B> boxes;
   std::vector<double> tris;      // 9 doubles per triangle, in the kinbody frame
   // <orcdchomp><spheres> of the kinbody (src/orcdchomp_kdata.cpp:79-94), in its own frame (one link): what create
   // reads from a body the robot holds (src/orcdchomp_mod.cpp:2173-2211); `link` is unused
   std::vector<Robot::Sphere> spheres;
};

// makes `device` the calling thread's current HIP device for the lifetime of the object
// (every entry point of a batch asserts its device: several modules, or the shards of one batch,
// may live on different GPUs of the node inside one process)
class DeviceGuard
{
public:
   explicit DeviceGuard(int device);
   ~DeviceGuard();
   DeviceGuard(const DeviceGuard &) = delete;
   DeviceGuard & operator=(const DeviceGuard &) = delete;
private:
   int prev_ = -1;
   bool changed_ = false;
};

// device memory released on the device it was allocated on
std::shared_ptr<void> device_buffer(int device, size_t bytes);

struct Sdf                        // struct sdf, src/orcdchomp_mod.cpp:148-153
{
   std::string kinbody_name;
   Pose pose;                     // grid wrt kinbody frame
   Grid grid;
   // device copies per device ordinal, created on demand; batches that read a copy share its
   // ownership, so removefield while a run exists does not pull the cells from under it
   std::map<int, std::shared_ptr<void>> dev64, dev32;
};

// a TSR hard constraint on every moving point (`con_tsr all ...` or `everyn_tsr`; struct tsr,
// src/orcdchomp_mod.h:80-87, struct run_contsr, src/orcdchomp_mod.cpp:873-885)
struct TsrSpec
{
   int ee_link = -1;
   Pose tool;                 // end effector in the link frame (identity for `link NAME`)
   Pose T0w, Twe;
   double Bw[6][2];
   int point = -1;            // -1: every moving point (`con_tsr all`, `everyn_tsr`); >= 0: that moving point only (`start_tsr`: 0)
};

struct BatchParams
{
   std::vector<TsrSpec> tsrs; // in the reference's order of addition: start_tsr, everyn_tsr, then the con_tsrs (mod.cpp:2570-2612)
   int free_start = 0;        // `start_tsr`: the start point is a variable (m = n_points - 1, no start boundary in the metric)
   int n_points = 101;
   int floating_base = 0;
   double lambda = 10.0;
   int derivative = 1;
   int use_momentum = 0;
   int use_hmc = 0;
   double hmc_resample_lambda = 0.02;
   double epsilon = 0.1, epsilon_self = 0.04, obs_factor = 200.0, obs_factor_self = 10.0;
   int precision = 64;
   int workgroup_threads = 0;   // 0: the module's setting (orc_set_workgroup_threads); `create` asks for 512 for its single run
   int workgroups_per_cu = 0;   // 0: the module's setting (orc_set_workgroups_per_cu)
};

class Module;

// a contiguous block of the runs of a batch on ONE device: n_runs independent runs sharing robot,
// fields and parameters (struct run, src/orcdchomp_mod.cpp:887-966, once per run in the reference)
class BatchShard
{
public:
   BatchShard(Module * mod, int device, hipStream_t stream, const Robot & robot, const BatchParams & p, int n_runs,
      const double * starts, const double * goals, const double * basegoals, const unsigned int * seeds);
   ~BatchShard();
   BatchShard(const BatchShard &) = delete;
   BatchShard & operator=(const BatchShard &) = delete;
   // iterations [iter_begin, iter_begin + n_iter) of an iterate call (r->iter restarts at 0 in every
   // call, src/orcdchomp_mod.cpp:2752: the hmc schedule compares against it), then the cost-only pass
   // `carry`: the launch continues an iterate call (runs that left their limits earlier in the call stay out)
   void iterate_async(int n_iter, int iter_begin = 0, bool final_eval = true, bool carry = false);
   void sync_begin(double * costs_out, int * status_out, int * iters_out);   // enqueue the copies
   void sync_end();                                                          // wait for them
   void gettraj(double * out);
   void get_plan(double out[8]) const;      // kernel variant bits, threads per workgroup, LDS bytes, tile, solve mode, workgroups per CU, tiles, lanes per waypoint
   void get_state(const std::string & which, double * out);
   void get_trace(double * out);
   void set_noise(const double * noise, int n_blocks);
   void set_traj(const double * traj);          // [n_runs][n_points][n] host -> device (warm start)
   // first contact of every run's trajectory with a field, on the device (Module::batch_collision_verdict plans the samples)
   void collision_verdict(const std::vector<int> & offs, const std::vector<int> & seg, const std::vector<double> & u,
                          const std::vector<int> & pairs, const std::vector<double> & pair_rsum, const std::vector<double> & inact_pos,
                          unsigned long long * key_out, double * depth_out);
   void get_phase_cycles(long long * out);   // [n_runs][8], diagnostics (ORC_PHASE_TIMERS=1)
   // kernel timing: completed event pairs are added to the module's totals (all of them when `wait`)
   void harvest_events(bool wait);

   int n_runs, n_points, n, m;
   BatchParams params;
   int last_n_iter = 0;
   int device;
   std::string robot_name;
   std::vector<int> adofindices;
   std::vector<int> device_sphere_order;    // XML index of device sphere k
   std::vector<int> slot_xml;               // XML index of the sphere in lane/slot q of the active block, -1: empty
private:
   template <typename real> void build_device(const Robot & robot);
   template <typename real> void launch(int n_iter, bool final_eval, bool carry);
   void plan_hmc(int iter_begin, int iter_end);
   int hmc_room(int n_iter) const;
   void hmc_reserve(int cap, bool pending_work);
   void construct(const Robot & robot, const double * starts, const double * goals, const double * basegoals,
      const unsigned int * seeds);
   void release();                  // frees every device buffer (destructor and failed construction)
   Module * mod_;
   hipStream_t stream_ = nullptr;   // the stream all work of this shard is issued on
   std::vector<std::shared_ptr<void>> sdf_refs_;   // the field copies the device descriptors point at
   std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_events_;
   Metric metric_;
   // device buffers (typed by params.precision)
   void * d_model_ = nullptr; void * d_sdfs_ = nullptr; void * d_sdfc_ = nullptr;
   void * d_traj_ = nullptr; void * d_AG_ = nullptr; void * d_G_ = nullptr; void * d_Gcost_ = nullptr;
   double * d_costs_ = nullptr; double * d_trace_ = nullptr; size_t trace_cap_ = 0;
   int * d_status_ = nullptr; int * d_iters_done_ = nullptr; int * d_leap_ = nullptr; long long * d_phase_ = nullptr;
   void * d_Aband_ = nullptr; void * d_beta_s_ = nullptr; void * d_beta_g_ = nullptr; void * d_metric64_ = nullptr;
   void * d_pcr_ = nullptr; void * d_Ainv_ = nullptr; void * d_jl_lo_ = nullptr; void * d_jl_hi_ = nullptr;
   // TSR hard constraints (csrc/tsr.h): the device copies of the constraints, the per-run workspace
   void * d_tsrs_ = nullptr; void * d_tsr_ws_ = nullptr; int * d_tsr_err_ = nullptr;
   int n_tsrs_ = 0, cons_k_ = 0; size_t tsr_ws_stride_ = 0;
   int * d_hmc_iters_ = nullptr; void * d_noise_ = nullptr; size_t hmc_cap_iters_ = 0, noise_cap_ = 0;
   int max_resamples_ = 0;
   hipEvent_t ev_plan_[2] = { nullptr, nullptr };   // iterate stream -> plan stream -> iterate stream
   int overflow_host_ = 0; bool overflow_armed_ = false;   // the plan's overflow flag, read (and cleared) with the results of a call
   bool plan_shared_ = false;                               // d_noise_ / d_hmc_iters_ are the module's buffers of this shard's stream
   bool unusable_ = false;                                  // set by an overflow: the runs' schedules were cut short, the batch has to be created again
   bool debug_state_ = false;   // ORC_DEBUG_STATE=1: keep the last gradient readable (get_state "G")
   int n_sdfs_ = 0;
   int n_tiles_ = 1, tile_first_ = 0, tile_rest_ = 0;   // tiles of an iteration: the first of tile_first_ moving waypoints, the others of tile_rest_
   ModelScalars ms_ = {};             // the device model's scalars (carried in the kernarg block)
   int Sa_real_ = 0;                  // active spheres
   int tsr_blocks_ = 0;               // (constraint, point) blocks of the TSR system
   int tsr_kmax_ = 0;                 // most constrained rows on one point
   int nj_ = 0, Sa_ = 0, S_ = 0;      // optimized joints; lanes of the active sphere block; lanes + inactive spheres
   int tile_m_ = 0;
   int block_ = 256;                  // threads per workgroup of the iterate kernel (256 or 192)
   int pcr_in_lds_ = 0;
   int tree_ = 0;
   bool pairs_latency_shape_ = false;      // the pair-list family's 512-thread kernels exist for this robot and precision (fp64 chains)
   int pair_entries_ = 0;             // entries of the staged self-collision pair list (rounds x 32; 0: the kernel family does not use one)
   int pcr_rows_ = 0, pcr_sym_ = 0, solve_mode_ = 0, ag_in_lds_ = 1, g_in_lds_ = 1, t_in_lds_ = 1, lds_flags_ = 0, GS_ = 0;
   size_t lds_bytes_ = 0;
   std::vector<double> jl_lo_, jl_hi_;
   // hmc host state per run (src/orcdchomp_mod.cpp:948-952)
   std::vector<GslRng> rng_;
   // ... or, for large batches, on the device (hmc_kernels.hip): mt19937 state [625][n_runs], next resample iteration [n_runs]
   bool hmc_on_device_ = false;
   uint32_t * d_mt_ = nullptr; uint32_t * d_mt_bak_ = nullptr; int * d_hmc_next_ = nullptr; int * d_hmc_next_bak_ = nullptr; int * d_overflow_ = nullptr;
   std::vector<int> hmc_resample_iter_;
   std::vector<double> ext_noise_; int ext_noise_blocks_ = 0;
   std::vector<int> ext_noise_used_;   // caller-supplied blocks consumed by the current iterate call, per run
   int stagger_mode_ = 0, stagger_sleeps_ = 10, lim_generic_ = 0;
};

// A batch as the boundary sees it: its runs are cut into contiguous blocks, one BatchShard per
// entry of the module's device list (no collective, the caller's arrays are the
// gather).  One device: one shard.
class Batch
{
public:
   Batch(Module * mod, const std::vector<int> & devices, const Robot & robot, const BatchParams & p, int n_runs,
      const double * starts, const double * goals, const double * basegoals, const unsigned int * seeds);
   ~Batch();
   void iterate_async(int n_iter, int iter_begin = 0, bool final_eval = true, bool carry = false);
   void sync(double * costs_out, int * status_out, int * iters_out = nullptr);
   void gettraj(double * out);
   void get_plan(double out[8]);            // the plan of the first shard (all shards of a batch plan alike)
   void get_state(const std::string & which, double * out);
   void get_trace(double * out);
   void set_noise(const double * noise, int n_blocks);
   void set_traj(const double * traj);
   void collision_verdict(const std::vector<int> & offs, const std::vector<int> & seg, const std::vector<double> & u,
                          const std::vector<int> & pairs, const std::vector<double> & pair_rsum, const std::vector<double> & inact_pos,
                          unsigned long long * key_out, double * depth_out);
   void get_phase_cycles(long long * out);
   // the per-iteration log of create's dat_filename (src/orcdchomp_mod.cpp:2306-2310, 2811-2818)
   void open_dat(const std::string & pattern);
   void write_dat(int iter_begin, int n_iter, const int * iters_done, double t_begin, double t_end);

   int n_runs, n_points, n, m;
   BatchParams params;
   int last_n_iter = 0;
   std::string robot_name;
   std::vector<int> adofindices;
   std::vector<int> device_sphere_order;
   std::vector<int> slot_xml;
   std::vector<Robot::Sphere> run_spheres;   // the spheres create collected (robot + held bodies): what the XML indices count through
   std::vector<unsigned char> run_self_excl; // [n][n] pairs of them the re-check's self-collision leg never tests (Robot::run_self_pairs_excluded, taken at create)
   std::vector<std::unique_ptr<BatchShard>> shards;
   std::vector<int> offs;            // first run of every shard, then n_runs
   bool has_dat() const { return !dat_.empty(); }
private:
   void for_shards(const std::function<void(size_t)> & body, bool threads);
   std::vector<FILE *> dat_;         // one per run (a single run: the reference's fp_dat)
};

class Module
{
public:
   explicit Module(int device);
   explicit Module(const std::vector<int> & devices);     // batches are sharded over these (repeats allowed)
   ~Module();
   // the SendCommand surface (src/orcdchomp_mod.h:58-66); throws std::runtime_error
   // with the reference's message strings
   std::string send_command(const std::string & cmd);

   // environment stand-ins
   void add_robot(const Robot & r);
   Robot & robot(const std::string & name);
   void add_kinbody(const KinBody & k);
   KinBody & kinbody(const std::string & name);
   bool has_body(const std::string & name) const;
   Pose body_transform(const std::string & name) const;   // robot or kinbody (a held kinbody: where its link carries it now)
   // RobotBase::Grab(body, link) / Release(body) / ReleaseAllGrabbed()
   void grab(const std::string & robot, const std::string & body, int link);
   void release(const std::string & robot, const std::string & body);
   void note_grab_contacts(Robot & r, Robot::Grab & g);      // fills touch_link / touch_body from the robot's state now
   void refresh_grab_contacts(const std::string & body);     // ... again, for a body that is held (its spheres were redefined)
   void set_kinbody_transform(const std::string & body, const Pose & pose);      // (a held body is re-anchored to its link)
   void release_all(const std::string & robot);
   // the robot as create collects its spheres (src/orcdchomp_mod.cpp:2148-2300): its own in XML order, then those of
   // every held body in GetGrabbed() order, each on the link that holds the body at T_w_rlink^-1 o T_w_klink o pos
   Robot robot_for_run(const std::string & name);

   // fields
   void add_sdf(const std::string & kinbody, const Grid & sdf, const Pose & pose_kinbody_gsdf);
   Sdf * find_sdf(const std::string & kinbody);
   std::vector<std::unique_ptr<Sdf>> sdfs;

   // lane placement of a robot's active spheres (place_spheres_on_row): a pure function of the robot
   // (geometry, limits, frozen dof values), the active dofs, floating base and epsilon_self
   std::map<std::string, std::vector<int>> placement_cache;
   // the shards of a batch are built on host threads of their own (Batch::Batch): the placement cache and the fields'
   // device copies (Sdf::dev64 / dev32) are taken under this
   std::recursive_mutex env_mutex;

   // batches
   int create_batch(const std::string & robot, const BatchParams & p, int n_runs,
      const double * starts, const double * goals, const double * basegoals, const unsigned int * seeds,
      const std::vector<int> * devices_override = nullptr);
   Batch & batch(int id);
   void destroy_batch(int id);
   // collision verdict of all runs of a batch (gettraj's re-check, batched on the device): per run
   // collides (0/1), time of the first contact on the retimed trajectory, XML sphere, field, depth
   void batch_collision_verdict(int id, int * collides, double * time, int * sphere, int * field, double * depth, bool self_check = true);

   hipStream_t stream = nullptr;     // orc_set_stream: the stream of the first device's work (NULL: its default stream)
   int device;                       // first entry of `devices`
   std::vector<int> devices;
   // optional pool of streams per device: shards are bound round-robin to one of them at creation so
   // that independent batches overlap on the GPU (the tail of one launch fills with the next)
   std::map<int, std::vector<hipStream_t>> stream_pool;
   std::map<int, size_t> next_pool_stream;
   int num_streams = 0;
   int workgroup_threads = 0;       // 0: the planner's choice; 192 or 256: the workgroup shape of every batch created from now on
   int workgroups_per_cu = 0;       // 0: the kernels' own register budget; 4: four 256-thread workgroups per CU where a kernel is built for it
   void set_num_streams(int n);
   hipStream_t pick_stream(int device, bool distinct);
   // a high-priority stream per device for the hmc plan of a call (hmc_kernels.hip): its wavefronts are dispatched ahead
   // of the iterate launches queued on the other streams instead of behind them
   hipStream_t plan_stream(int device);
   // The momentum-resampling plan of an iterate call (noise [n_runs][cap][m n], resample iterations [n_runs][cap]) is written and
   // read inside that one call, so the batches that run on one stream share one pair of buffers: the stream orders their
   // calls.  (A buffer per batch held 1.8 GB for every config-4 batch a caller had created ahead of time.)
   struct PlanBuffers { void * noise = nullptr; size_t noise_bytes = 0; int * iters = nullptr; size_t iters_count = 0; std::mutex enqueue; };
   PlanBuffers & plan_buffers(int device, hipStream_t stream);
   // kernel timing (HIP events on the shards' streams), harvested from the shards
   void time_collect();
   double kernel_ms_total = 0.0;
   int kernel_launches = 0;
   // The pool of timing events and the totals are shared by all shards, and the shards of one batch may
   // launch from host threads of their own (Batch::for_shards): everything below takes timing_mutex_.
   hipEvent_t acquire_event(int device);                       // a pooled event of `device`, or a new one (the device must be current)
   void release_event(int device, hipEvent_t ev);
   void add_kernel_time(double ms);
   std::string last_error;
   std::string last_reply;
   std::string last_collision_details;   // what the reference logs with RAVELOG_ERROR in gettraj

private:
   std::string cmd_computedistancefield(const std::vector<std::string> & argv);
   std::string cmd_addfield_fromobsarray(const std::vector<std::string> & argv);
   std::string cmd_removefield(const std::vector<std::string> & argv);
   std::string cmd_create(const std::vector<std::string> & argv, bool batch);
   std::string cmd_iterate(const std::vector<std::string> & argv, bool batch);
   std::string cmd_gettraj(const std::vector<std::string> & argv, bool batch);
   std::string cmd_destroy(const std::vector<std::string> & argv);
   std::map<std::string, Robot> robots_;
   std::map<std::string, KinBody> kinbodies_;
   std::map<int, std::unique_ptr<Batch>> batches_;
   int next_batch_id_ = 1;
   std::map<int, std::vector<hipEvent_t>> event_pool_;
   std::map<int, hipStream_t> plan_streams_;
   std::map<std::pair<int, hipStream_t>, PlanBuffers> plan_buffers_;
   std::mutex timing_mutex_;
};

void hip_check(hipError_t e, const char * what);
int count_int_conversions(const std::string & pattern);   // integer conversions of a printf pattern, -1: it holds another kind

} // namespace orc
