// module.cpp -- environment stand-ins, SDF commands, command grammar.
// Reference: src/orcdchomp_mod.cpp (commands), src/orcwrap.cpp (argv adaptor).
#include "module.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <limits>
#include <sstream>
#include <stdexcept>

hipError_t orc_sdf_from_occupancy_device(const double * occ, double * sdf_out, const int sizes[3], const double lengths[3],
   hipStream_t st);
hipError_t orc_sdf_build_device(const int sizes[3], const double lengths[3], const double grid_xform[12], const double grid_pose[7],
   double cube_extent, int n_boxes, const double * boxes, int n_tris, const double * tris, double * sdf_out, hipStream_t st);

namespace orc {

// occupancy -> signed distance field: on the GPU for large grids (or ORC_SDF_DEVICE=1), on the host
// otherwise (ORC_SDF_DEVICE=0 forces the host); both are bit-identical
static void bin_sdf_any(const Grid & occ, Grid & sdf, hipStream_t st)
{
   const char * env = getenv("ORC_SDF_DEVICE");
   const bool on_device = env ? (atoi(env) != 0) : (occ.ncells() >= (size_t) 1 << 18);
   if (!on_device) { grid_bin_sdf(occ, sdf); return; }
   sdf = occ;
   hip_check(orc_sdf_from_occupancy_device(occ.data.data(), sdf.data.data(), occ.sizes, occ.lengths, st), "sdf build");
}

void hip_check(hipError_t e, const char * what)
{
   if (e != hipSuccess)
      throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}

// ================================================================ Robot ===
bool Robot::does_affect(int dof, int link) const
{
   for (int li=link; li>=0; li=parent[li])
      if (joint_type[li] != 0 && dof_index[li] == dof) return true;
   return false;
}

std::vector<unsigned char> Robot::self_pairs_excluded() const
{
   const int nl = n_links;
   std::vector<unsigned char> excl((size_t) nl * nl, 0);
   for (int a=0; a<nl; a++)
   {
      excl[(size_t) a*nl + a] = 1;
      if (parent[a] >= 0) { excl[(size_t) a*nl + parent[a]] = 1; excl[(size_t) parent[a]*nl + a] = 1; }
   }
   for (const auto & pr : adjacent) { excl[(size_t) pr.first*nl + pr.second] = 1; excl[(size_t) pr.second*nl + pr.first] = 1; }
   std::vector<Xform> frames;
   fk(Pose(), std::vector<double>(n_dof > 0 ? n_dof : 1, 0.0), frames);
   std::vector<double> pw(spheres.size() * 3);
   for (size_t a=0; a<spheres.size(); a++)
   {
      double r[3];
      mat3_vec(frames[spheres[a].link].R, spheres[a].pos, r);
      for (int k=0; k<3; k++) pw[a*3+k] = r[k] + frames[spheres[a].link].t[k];
   }
   for (size_t a=0; a<spheres.size(); a++)
      for (size_t b=a+1; b<spheres.size(); b++)
      {
         double d2 = 0.0;
         for (int k=0; k<3; k++) { const double d = pw[a*3+k] - pw[b*3+k]; d2 += d*d; }
         if (std::sqrt(d2) - (spheres[a].radius + spheres[b].radius) < 0.0)
         {
            excl[(size_t) spheres[a].link*nl + spheres[b].link] = 1;
            excl[(size_t) spheres[b].link*nl + spheres[a].link] = 1;
         }
      }
   return excl;
}

// The re-check's self-collision leg for a robot that holds bodies.  OpenRAVE's CheckSelfCollision tests the robot's links
// against each other (minus adjacent links: self_pairs_excluded, from the ROBOT's own spheres) and every grabbed body against the
// links it was NOT touching when it was grabbed (the grabbing link is always left out).  For the sphere model: a pair of the
// robot's own spheres follows the link rule; two spheres of one held body are one rigid body; a held body's sphere against
// anything else is left out when both ride on the same link, or when the body overlapped that link's own spheres (or that other
// body) at the moment of the grab (Grab::touch_link / touch_body, recorded by Module::grab -- not in the configuration the run
// is created in: a body that has come to touch a link since the grab is reported against it).
std::vector<unsigned char> Robot::run_self_pairs_excluded(int n_own) const
{
   const int ns = (int) spheres.size();
   std::vector<unsigned char> excl((size_t) ns * ns, 0);
   Robot own = *this;
   own.spheres.resize(n_own);
   const std::vector<unsigned char> link_excl = own.self_pairs_excluded();
   auto body_touches_link = [&](int body, int link) {
      const Grab & g = grabbed[body - 1];
      return link == g.link || (link < (int) g.touch_link.size() && g.touch_link[link] != 0);
   };
   auto bodies_touch = [&](int ba, int bb) {
      const Grab & ga = grabbed[ba - 1], & gb = grabbed[bb - 1];
      for (const std::string & nm : ga.touch_body) if (nm == gb.body) return true;
      for (const std::string & nm : gb.touch_body) if (nm == ga.body) return true;
      return false;
   };
   for (int a=0; a<ns; a++)
      for (int b=a+1; b<ns; b++)
      {
         const int ba = spheres[a].body, bb = spheres[b].body;
         bool ex;
         if (ba == 0 && bb == 0) ex = link_excl[(size_t) spheres[a].link * n_links + spheres[b].link] != 0;
         else if (ba == bb) ex = true;
         else if (ba == 0) ex = body_touches_link(bb, spheres[a].link);
         else if (bb == 0) ex = body_touches_link(ba, spheres[b].link);
         else ex = bodies_touch(ba, bb);
         excl[(size_t) a*ns + b] = excl[(size_t) b*ns + a] = ex ? 1 : 0;
      }
   for (int a=0; a<ns; a++) excl[(size_t) a*ns + a] = 1;
   return excl;
}

void Robot::fk(const Pose & base, const std::vector<double> & q, std::vector<Xform> & frames) const
{
   frames.resize(n_links);
   const Xform xb = xform_from_pose(base);
   for (int li=0; li<n_links; li++)
   {
      const Xform & from = (parent[li] < 0) ? xb : frames[parent[li]];
      Xform xj = xform_mul(from, xform_from_pose(pose_parent_joint[li]));
      if (joint_type[li] == 1)
      {
         Xform rot; rot.R = axis_angle(&axis[3*li], q[dof_index[li]]); rot.t[0] = rot.t[1] = rot.t[2] = 0.0;
         xj = xform_mul(xj, rot);
      }
      else if (joint_type[li] == 2)
      {
         double aw[3];
         mat3_vec(xj.R, &axis[3*li], aw);
         for (int k=0; k<3; k++) xj.t[k] += q[dof_index[li]] * aw[k];
      }
      frames[li] = xj;
   }
}

// =============================================================== Module ===
DeviceGuard::DeviceGuard(int device)
{
   if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
   if (prev_ != device)
   {
      hip_check(hipSetDevice(device), "hipSetDevice");
      changed_ = true;
   }
}

DeviceGuard::~DeviceGuard()
{
   if (changed_ && prev_ >= 0) (void) hipSetDevice(prev_);
}

std::shared_ptr<void> device_buffer(int device, size_t bytes)
{
   DeviceGuard guard(device);
   void * p = nullptr;
   hip_check(hipMalloc(&p, bytes ? bytes : 1), "hipMalloc");
   return std::shared_ptr<void>(p, [device](void * q) {
      int prev = -1;
      const bool have = hipGetDevice(&prev) == hipSuccess;
      (void) hipSetDevice(device);
      (void) hipFree(q);
      if (have && prev != device) (void) hipSetDevice(prev);
   });
}

Module::Module(int dev) : Module(std::vector<int>(1, dev)) {}

Module::Module(const std::vector<int> & devs) : device(devs.empty() ? 0 : devs[0]), devices(devs)
{
   int count = 0;
   hipError_t e = hipGetDeviceCount(&count);
   if (e != hipSuccess || count <= 0)
      throw std::runtime_error("orcdchomp_amd: no HIP device available (the MI355X path has no CPU fallback)");
   if (devices.empty()) throw std::runtime_error("orcdchomp_amd: empty device list");
   for (int d : devices) if (d < 0 || d >= count) throw std::runtime_error("orcdchomp_amd: bad device ordinal");
   hip_check(hipSetDevice(device), "hipSetDevice");
}

Module::~Module()
{
   batches_.clear();
   sdfs.clear();
   for (auto & kv : event_pool_)
   {
      DeviceGuard guard(kv.first);
      for (hipEvent_t ev : kv.second) (void) hipEventDestroy(ev);
   }
   for (auto & kv : stream_pool)
   {
      DeviceGuard guard(kv.first);
      for (hipStream_t st : kv.second) (void) hipStreamDestroy(st);
   }
   for (auto & kv : plan_streams_)
   {
      DeviceGuard guard(kv.first);
      (void) hipStreamDestroy(kv.second);
   }
   for (auto & kv : plan_buffers_)
   {
      DeviceGuard guard(kv.first.first);
      (void) hipFree(kv.second.noise); (void) hipFree(kv.second.iters);
   }
}

hipEvent_t Module::acquire_event(int dev)
{
   {
      std::lock_guard<std::mutex> lock(timing_mutex_);
      std::vector<hipEvent_t> & pool = event_pool_[dev];
      if (!pool.empty()) { hipEvent_t ev = pool.back(); pool.pop_back(); return ev; }
   }
   hipEvent_t ev;
   hip_check(hipEventCreate(&ev), "hipEventCreate");
   return ev;
}

void Module::release_event(int dev, hipEvent_t ev)
{
   std::lock_guard<std::mutex> lock(timing_mutex_);
   event_pool_[dev].push_back(ev);
}

void Module::add_kernel_time(double ms)
{
   std::lock_guard<std::mutex> lock(timing_mutex_);
   kernel_ms_total += ms;
   kernel_launches++;
}

// a pool of n streams on every device of the module; live batches hold the streams they were bound
// to, so the pool only changes while no batch exists
void Module::set_num_streams(int n)
{
   if (!batches_.empty())
      throw std::runtime_error("orc_set_num_streams: destroy the existing batches first (they hold the pool's streams)");
   for (auto & kv : stream_pool)
   {
      DeviceGuard guard(kv.first);
      for (hipStream_t st : kv.second) (void) hipStreamDestroy(st);
   }
   stream_pool.clear();
   next_pool_stream.clear();
   num_streams = n;
}

// the stream a new shard on `dev` is bound to: round-robin over the device's pool when there is one
// (created on first use), else the module's stream on the first device and the default stream
// elsewhere.  `distinct`: the shard shares its device with another shard of the same batch and must
// not queue behind it.
hipStream_t Module::pick_stream(int dev, bool distinct)
{
   int want = num_streams;
   if (distinct && want < 2)
   {
      int same = 0;
      for (int d : devices) if (d == dev) same++;
      want = same > 2 ? same : 2;
   }
   if (want <= 0) return dev == device ? stream : nullptr;
   std::vector<hipStream_t> & pool = stream_pool[dev];
   if ((int) pool.size() < want)
   {
      DeviceGuard guard(dev);
      while ((int) pool.size() < want)
      {
         hipStream_t st;
         hip_check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
         pool.push_back(st);
      }
   }
   return pool[next_pool_stream[dev]++ % pool.size()];
}

hipStream_t Module::plan_stream(int dev)
{
   std::lock_guard<std::mutex> lock(timing_mutex_);
   auto it = plan_streams_.find(dev);
   if (it != plan_streams_.end()) return it->second;
   DeviceGuard guard(dev);
   int least = 0, greatest = 0;
   hip_check(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
   hipStream_t st;
   hip_check(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest), "hipStreamCreateWithPriority");
   plan_streams_[dev] = st;
   return st;
}

Module::PlanBuffers & Module::plan_buffers(int dev, hipStream_t stream)
{
   std::lock_guard<std::mutex> lock(timing_mutex_);
   return plan_buffers_[std::make_pair(dev, stream)];      // (references to a map's elements stay valid)
}

void Module::time_collect()
{
   for (auto & kv : batches_)
      for (auto & sh : kv.second->shards) sh->harvest_events(false);
}

void Module::add_robot(const Robot & r)
{
   if (has_body(r.name)) throw std::runtime_error("a body with that name already exists!");
   robots_[r.name] = r;
}

Robot & Module::robot(const std::string & name)
{
   auto it = robots_.find(name);
   if (it == robots_.end()) throw std::runtime_error("Could not find robot with that name!");
   return it->second;
}

void Module::add_kinbody(const KinBody & k)
{
   if (has_body(k.name)) throw std::runtime_error("a body with that name already exists!");
   kinbodies_[k.name] = k;
}

KinBody & Module::kinbody(const std::string & name)
{
   auto it = kinbodies_.find(name);
   if (it == kinbodies_.end()) throw std::runtime_error("Could not find kinbody with that name!");
   return it->second;
}

bool Module::has_body(const std::string & name) const
{
   return robots_.count(name) || kinbodies_.count(name);
}

Pose Module::body_transform(const std::string & name) const
{
   auto r = robots_.find(name);
   if (r != robots_.end()) return r->second.transform;
   auto k = kinbodies_.find(name);
   if (k != kinbodies_.end())
   {
      // a held body moves with the link that holds it (OpenRAVE updates grabbed bodies with the robot's state)
      for (const auto & kv : robots_)
         for (const Robot::Grab & g : kv.second.grabbed)
            if (g.body == name)
            {
               std::vector<Xform> frames;
               kv.second.fk(kv.second.transform, kv.second.dof_values, frames);
               const Xform now = xform_mul(frames[g.link], g.rel);
               return pose_from_dR(now.t, now.R);
            }
      return k->second.transform;
   }
   throw std::runtime_error("KinBody " + name + " referenced by active signed distance field does not exist!\n");
}

void Module::grab(const std::string & rname, const std::string & body, int link)
{
   Robot & r = robot(rname);
   KinBody & k = kinbody(body);
   if (link < 0 || link >= r.n_links) throw std::runtime_error("grabbing link out of range!");
   for (const auto & kv : robots_)
      for (const Robot::Grab & g : kv.second.grabbed)
         if (g.body == body) throw std::runtime_error("that kinbody is already grabbed!");
   std::vector<Xform> frames;
   r.fk(r.transform, r.dof_values, frames);
   Robot::Grab g;
   g.body = body; g.link = link;
   g.rel = xform_mul(xform_inverse(frames[link]), xform_from_pose(k.transform));
   note_grab_contacts(r, g);
   r.grabbed.push_back(g);
}

// What a body overlaps at the moment it is grabbed (or re-anchored): the robot's links, by the robot's own <orcdchomp> spheres
// against the body's, and the bodies the robot holds already.  RobotBase::Grab records the links in collision with the body and
// CheckSelfCollision ignores them afterwards (third party; src/orcdchomp_mod.cpp:2998-2999 calls it).
void Module::note_grab_contacts(Robot & r, Robot::Grab & g)
{
   g.touch_link.assign(r.n_links, 0);
   g.touch_body.clear();
   const KinBody & k = kinbody(g.body);
   std::vector<Xform> frames;
   r.fk(r.transform, r.dof_values, frames);
   auto world = [&](const Xform & x, const double pos[3], double out[3]) {
      mat3_vec(x.R, pos, out);
      for (int q=0; q<3; q++) out[q] += x.t[q];
   };
   const Xform xb = xform_mul(frames[g.link], g.rel);
   std::vector<double> pb(k.spheres.size() * 3);
   for (size_t a=0; a<k.spheres.size(); a++) world(xb, k.spheres[a].pos, &pb[a*3]);
   auto overlap = [](const double * p, double rp, const double * q, double rq) {
      double d2 = 0.0;
      for (int c=0; c<3; c++) { const double d = p[c] - q[c]; d2 += d*d; }
      return std::sqrt(d2) - (rp + rq) < 0.0;
   };
   for (const Robot::Sphere & sp : r.spheres)
   {
      double pw[3];
      world(frames[sp.link], sp.pos, pw);
      for (size_t a=0; a<k.spheres.size(); a++)
         if (overlap(pw, sp.radius, &pb[a*3], k.spheres[a].radius)) { g.touch_link[sp.link] = 1; break; }
   }
   for (const Robot::Grab & other : r.grabbed)
   {
      if (other.body == g.body) continue;
      const KinBody & ko = kinbody(other.body);
      const Xform xo = xform_mul(frames[other.link], other.rel);
      bool hit = false;
      for (size_t a=0; a<ko.spheres.size() && !hit; a++)
      {
         double po[3];
         world(xo, ko.spheres[a].pos, po);
         for (size_t c=0; c<k.spheres.size() && !hit; c++)
            if (overlap(po, ko.spheres[a].radius, &pb[c*3], k.spheres[c].radius)) hit = true;
      }
      if (hit) g.touch_body.push_back(other.body);
   }
}

// SetTransform of a kinbody.  A body the robot holds is moved where the caller says and rides with its link from THERE
// (the grab's relative transform is taken anew): a binding that copies every body's pose before a command passes a held
// body's current pose, which changes nothing.
void Module::set_kinbody_transform(const std::string & body, const Pose & pose)
{
   KinBody & k = kinbody(body);
   k.transform = pose;
   for (auto & kv : robots_)
      for (Robot::Grab & g : kv.second.grabbed)
         if (g.body == body)
         {
            std::vector<Xform> frames;
            kv.second.fk(kv.second.transform, kv.second.dof_values, frames);
            g.rel = xform_mul(xform_inverse(frames[g.link]), xform_from_pose(pose));
            note_grab_contacts(kv.second, g);      // (a new anchoring is a new grab: what it touches is taken from here)
         }
}

void Module::refresh_grab_contacts(const std::string & body)
{
   for (auto & kv : robots_)
      for (Robot::Grab & g : kv.second.grabbed)
         if (g.body == body) note_grab_contacts(kv.second, g);
}

void Module::release(const std::string & rname, const std::string & body)
{
   Robot & r = robot(rname);
   for (size_t i=0; i<r.grabbed.size(); i++)
      if (r.grabbed[i].body == body)
      {
         kinbody(body).transform = body_transform(body);      // the body stays where the link left it
         r.grabbed.erase(r.grabbed.begin() + i);
         return;
      }
   throw std::runtime_error("the robot is not grabbing that kinbody!");
}

void Module::release_all(const std::string & rname)
{
   Robot & r = robot(rname);
   while (!r.grabbed.empty()) release(rname, r.grabbed.back().body);
}

Robot Module::robot_for_run(const std::string & rname)
{
   Robot eff = robot(rname);
   // a body without spheres, the robot itself included (src/orcdchomp_mod.cpp:2262-2263)
   if (eff.spheres.empty()) throw std::runtime_error("no spheres! kinbody does not have a <orcdchomp> tag defined?");
   int body_index = 0;
   for (const Robot::Grab & g : eff.grabbed)
   {
      body_index++;
      const KinBody & k = kinbody(g.body);
      if (k.spheres.empty()) throw std::runtime_error("no spheres! kinbody does not have a <orcdchomp> tag defined?");
      for (const Robot::Sphere & ks : k.spheres)
      {
         // T_w_rlink^-1 o T_w_klink o pos (mod.cpp:2200-2208); the held body is rigid with its link, so the product of
         // the first two is what the grab recorded
         Robot::Sphere sp;
         sp.link = g.link; sp.radius = ks.radius; sp.body = body_index;
         mat3_vec(g.rel.R, ks.pos, sp.pos);
         for (int q=0; q<3; q++) sp.pos[q] += g.rel.t[q];
         eff.spheres.push_back(sp);
      }
   }
   return eff;
}

Sdf * Module::find_sdf(const std::string & kinbody)
{
   for (auto & s : sdfs) if (s->kinbody_name == kinbody) return s.get();
   return nullptr;
}

void Module::add_sdf(const std::string & kb, const Grid & g, const Pose & pose)
{
   if (find_sdf(kb)) throw std::runtime_error("We already have an sdf for this kinbody!");
   for (int d=0; d<3; d++)
      if (g.sizes[d] < 2) throw std::runtime_error("sdf grids need at least 2 cells per dimension!");
   std::unique_ptr<Sdf> s(new Sdf);
   s->kinbody_name = kb;
   s->pose = pose;
   s->grid = g;
   sdfs.push_back(std::move(s));
}

int Module::create_batch(const std::string & rname, const BatchParams & p, int n_runs,
   const double * starts, const double * goals, const double * basegoals, const unsigned int * seeds,
   const std::vector<int> * devices_override)
{
   const Robot r = robot_for_run(rname);
   if (devices_override)
   {
      int count = 0;
      hip_check(hipGetDeviceCount(&count), "hipGetDeviceCount");
      for (int d : *devices_override) if (d < 0 || d >= count) throw std::runtime_error("orcdchomp_amd: bad device ordinal");
   }
   std::unique_ptr<Batch> b(new Batch(this, devices_override ? *devices_override : devices, r, p, n_runs, starts, goals, basegoals, seeds));
   b->run_spheres = r.spheres;
   b->run_self_excl = r.run_self_pairs_excluded((int) robot(rname).spheres.size());
   const int id = next_batch_id_++;
   batches_[id] = std::move(b);
   return id;
}

Batch & Module::batch(int id)
{
   auto it = batches_.find(id);
   if (it == batches_.end()) throw std::runtime_error("you must pass a created run!");
   return *it->second;
}

void Module::destroy_batch(int id)
{
   auto it = batches_.find(id);
   if (it == batches_.end()) throw std::runtime_error("you must pass a created run!");
   batches_.erase(it);
}

// ------------------------------------------------------------ commands ---
namespace {

std::vector<double> parse_vector(const std::string & s)
{
   std::vector<double> out;
   for (const std::string & tok : shparse(s)) out.push_back(std::atof(tok.c_str()));
   return out;
}

void bad_arguments() { throw std::runtime_error("Bad arguments!"); }

void * parse_pointer(const std::string & s)
{
   void * p = nullptr;
   if (std::sscanf(s.c_str(), "%p", &p) != 1) return nullptr;
   return p;
}

// an OpenRAVE-style trajectory document holding the waypoints of one run
// (stands in for t->serialize(sout), src/orcdchomp_mod.cpp:3008).  One row per waypoint:
// the joint values followed by the deltatime to reach it from the previous waypoint.
std::string serialize_traj(const std::string & robot, const std::vector<int> & adofs,
   const double * traj, int n_points, int n, int col0, const std::vector<double> & deltatime)
{
   std::ostringstream o;
   const int nd = n - col0;
   o << std::setprecision(std::numeric_limits<double>::max_digits10);
   o << "<trajectory>\n<configuration>\n<group name=\"joint_values " << robot;
   for (int a : adofs) o << " " << a;
   o << "\" offset=\"0\" dof=\"" << nd << "\" interpolation=\"linear\"/>\n";
   o << "<group name=\"deltatime\" offset=\"" << nd << "\" dof=\"1\" interpolation=\"\"/>\n";
   if (col0)
   {
      // floating base: the second trajectory of src/orcdchomp_mod.cpp:2912-2956 merged in: the base pose of every
      // waypoint as `affine_transform <robot> <DOF_Transform>` (x y z, then the quaternion in OpenRAVE's order
      // w x y z) and its finite-difference velocities over the waypoint's deltatime as `affine_velocities`
      const int dof_transform = 1 | 2 | 4 | 32;      // OpenRAVE::DOF_XYZ | DOF_RotationQuat
      o << "<group name=\"affine_transform " << robot << " " << dof_transform << "\" offset=\"" << nd + 1 << "\" dof=\"7\" interpolation=\"linear\"/>\n";
      o << "<group name=\"affine_velocities " << robot << " " << dof_transform << "\" offset=\"" << nd + 8 << "\" dof=\"7\" interpolation=\"next\"/>\n";
   }
   o << "</configuration>\n<data count=\"" << n_points << "\">\n";
   static const int order[7] = { 0, 1, 2, 6, 3, 4, 5 };      // libcd x y z qx qy qz qw -> x y z qw qx qy qz
   for (int i=0; i<n_points; i++)
   {
      for (int j=col0; j<n; j++) o << traj[(size_t) i*n+j] << " ";
      o << deltatime[i];
      if (col0)
      {
         for (int k=0; k<7; k++) o << " " << traj[(size_t) i*n + order[k]];
         for (int k=0; k<7; k++)
            o << " " << ((i > 0) ? (traj[(size_t) i*n + order[k]] - traj[(size_t)(i-1)*n + order[k]]) / deltatime[i] : 0.0);
      }
      if (i+1 < n_points) o << " ";
   }
   o << "\n</data>\n</trajectory>\n";
   return o.str();
}

// LinearTrajectoryRetimer stand-in (reference: RetimeActiveDOFTrajectory(..., "LinearTrajectoryRetimer"),
// src/orcdchomp_mod.cpp:2905-2911): each segment is traversed at the largest constant velocity
// the dof velocity limits allow.
std::vector<double> retime_linear(const double * traj, int n_points, int n, int col0, const std::vector<double> & vmax)
{
   std::vector<double> dtm(n_points, 0.0);
   for (int i=1; i<n_points; i++)
      for (int j=col0; j<n; j++)
      {
         const double v = vmax[j-col0] > 0.0 ? vmax[j-col0] : 1.0;
         dtm[i] = std::max(dtm[i], std::fabs(traj[(size_t) i*n+j] - traj[(size_t)(i-1)*n+j]) / v);
      }
   return dtm;
}

// reads a trajectory document: the rows of its <data> block and where its groups sit in a row.  A document without
// a <configuration> block (the bare form the tests also pass) is joint values followed by one deltatime.
struct TrajDoc
{
   int count = 0, width = 0;
   std::vector<double> vals;                       // [count][width]
   int joint_off = 0, joint_dof = 0, dt_off = -1, base_off = -1;
};
bool parse_traj(const std::string & text, TrajDoc & doc)
{
   const size_t c0 = text.find("<data count=\"");
   if (c0 == std::string::npos) return false;
   doc.count = std::atoi(text.c_str() + c0 + 13);
   const size_t d0 = text.find('>', c0), d1 = text.find("</data>", c0);
   if (d0 == std::string::npos || d1 == std::string::npos || doc.count < 1) return false;
   std::istringstream is(text.substr(d0+1, d1-d0-1));
   double v;
   while (is >> v) doc.vals.push_back(v);
   if (doc.vals.empty() || doc.vals.size() % (size_t) doc.count) return false;
   doc.width = (int)(doc.vals.size() / doc.count);
   bool any_group = false;
   for (size_t g=text.find("<group name=\""); g!=std::string::npos && g<c0; g=text.find("<group name=\"", g+1))
   {
      const size_t n0 = g + 13, n1 = text.find('"', n0);
      const size_t o0 = text.find("offset=\"", n1), f0 = text.find("dof=\"", n1);
      if (n1 == std::string::npos || o0 == std::string::npos || f0 == std::string::npos) return false;
      const std::string name = text.substr(n0, n1 - n0);
      const int off = std::atoi(text.c_str() + o0 + 8), dof = std::atoi(text.c_str() + f0 + 5);
      if (off < 0 || dof < 1 || off + dof > doc.width) return false;
      any_group = true;
      if (name.compare(0, 12, "joint_values") == 0) { doc.joint_off = off; doc.joint_dof = dof; }
      else if (name == "deltatime") doc.dt_off = off;
      else if (name.compare(0, 16, "affine_transform") == 0 && dof == 7) doc.base_off = off;
   }
   if (!any_group) { doc.joint_off = 0; doc.joint_dof = doc.width - 1; doc.dt_off = doc.width - 1; }
   return doc.joint_dof >= 1 && doc.dt_off >= 0;
}

} // namespace

std::string Module::send_command(const std::string & cmd)
{
   // orcwrap_call (src/orcwrap.cpp:37-69): tokenise; argv[0] is the command name
   std::vector<std::string> argv = shparse(cmd);
   if (argv.empty()) throw std::runtime_error("empty command!");
   const std::string name = argv[0];
   if (name == "computedistancefield") return cmd_computedistancefield(argv);
   if (name == "addfield_fromobsarray") return cmd_addfield_fromobsarray(argv);
   if (name == "removefield") return cmd_removefield(argv);
   if (name == "create") return cmd_create(argv, false);
   if (name == "createbatch") return cmd_create(argv, true);
   if (name == "iterate") return cmd_iterate(argv, false);
   if (name == "iteratebatch") return cmd_iterate(argv, true);
   if (name == "gettraj") return cmd_gettraj(argv, false);
   if (name == "gettrajbatch") return cmd_gettraj(argv, true);
   if (name == "destroy") return cmd_destroy(argv);
   if (name == "viewspheres" || name == "viewfields")
      throw std::runtime_error("command " + name + " needs an OpenRAVE viewer and is not part of this build");
   throw std::runtime_error("unknown command " + name);
}

// src/orcdchomp_mod.cpp:297-589
std::string Module::cmd_computedistancefield(const std::vector<std::string> & argv)
{
   std::string kb;
   double cube_extent = 0.02, aabb_padding = 0.2;
   std::string cache_filename;
   bool have_cache = false, require_cache = false;
   const int argc = (int) argv.size();
   int i;
   for (i=1; i<argc; i++)
   {
      if (argv[i] == "kinbody" && i+1 < argc)
      {
         if (!kb.empty()) throw std::runtime_error("Only one kinbody can be passed!");
         kb = argv[++i];
         if (!has_body(kb)) throw std::runtime_error("Could not find kinbody with that name!");
      }
      else if (argv[i] == "aabb_padding" && i+1 < argc) aabb_padding = std::atof(argv[++i].c_str());
      else if (argv[i] == "cube_extent" && i+1 < argc) cube_extent = std::atof(argv[++i].c_str());
      else if (argv[i] == "cache_filename" && i+1 < argc) { cache_filename = argv[++i]; have_cache = true; }
      else if (argv[i] == "require_cache") require_cache = true;
      else break;
   }
   if (i < argc) bad_arguments();
   if (kb.empty()) throw std::runtime_error("Did not pass all required args!");
   if (find_sdf(kb)) throw std::runtime_error("We already have an sdf for this kinbody!");

   // AABB of the enabled geometry with the body at the world origin (mod.cpp:377-381, 88-140)
   double amin[3] = {0,0,0}, amax[3] = {0,0,0};
   bool init = false;
   if (kinbodies_.count(kb))
   {
      const KinBody & k = kinbodies_[kb];
      if (k.enabled) for (size_t v=0; v<k.tris.size()/3; v++)      // (a mesh: the box around its vertices)
         for (int r=0; r<3; r++)
         {
            const double c = k.tris[3*v + r];
            if (!init) { amin[r] = c; amax[r] = c; }
            else { if (amin[r] > c) amin[r] = c; if (amax[r] < c) amax[r] = c; }
            if (r == 2) init = true;
         }
      if (k.enabled) for (const KinBody::B & bx : k.boxes)
      {
         const Xform x = xform_from_pose(bx.pose);
         double ext[3];
         for (int r=0; r<3; r++)
            ext[r] = std::fabs(x.R.m[r*3+0])*bx.half[0] + std::fabs(x.R.m[r*3+1])*bx.half[1] + std::fabs(x.R.m[r*3+2])*bx.half[2];
         if (ext[0] == 0 && ext[1] == 0 && ext[2] == 0) continue;
         for (int r=0; r<3; r++)
         {
            const double lo = x.t[r] - ext[r], hi = x.t[r] + ext[r];
            if (!init) { amin[r] = lo; amax[r] = hi; }
            else { if (amin[r] > lo) amin[r] = lo; if (amax[r] < hi) amax[r] = hi; }
         }
         init = true;
      }
   }
   double apos[3], aext[3];
   for (int r=0; r<3; r++)
   {
      if (init) { apos[r] = 0.5 * (amin[r] + amax[r]); aext[r] = amax[r] - apos[r]; }
      else { apos[r] = 0.0; aext[r] = 0.0; }      // identity transform translation (mod.cpp:131-134)
   }

   Grid g;
   for (int r=0; r<3; r++)
   {
      g.sizes[r] = (int) std::ceil((aext[r] + aabb_padding) / cube_extent);      // mod.cpp:391
      g.lengths[r] = g.sizes[r] * 2.0 * cube_extent;                              // mod.cpp:403
   }
   for (int r=0; r<3; r++)
      if (g.sizes[r] < 2) throw std::runtime_error("Not enough memory for distance field!");
   Pose pose_gsdf;
   for (int r=0; r<3; r++) pose_gsdf.v[r] = apos[r] - 0.5 * g.lengths[r];         // mod.cpp:407-409
   g.data.assign(g.ncells(), 1.0);

   bool loaded = false;
   if (have_cache)
   {
      // raw doubles, C order, no header; validated by size only (mod.cpp:416-444)
      std::ifstream fp(cache_filename.c_str(), std::ios::binary | std::ios::ate);
      if (fp)
      {
         const std::streamoff sz = fp.tellg();
         if ((size_t) sz == g.ncells() * sizeof(double))
         {
            fp.seekg(0);
            fp.read((char *) g.data.data(), sz);
            loaded = (bool) fp;
         }
      }
   }
   if (!loaded)
   {
      if (require_cache) throw std::runtime_error("Field not found from cache, but require_cache flag set!");
      // occupancy by sweeping a cube of half-extent cube_extent over the cell centres
      // (mod.cpp:462-531; OpenRAVE's CheckCollision replaced by a box-box test)
      const Pose pose_world_gsdf = pose_compose(body_transform(kb), pose_gsdf);
      std::vector<Box> obstacles;
      std::vector<double> tris;                                            // 9 doubles per triangle, world coordinates
      for (auto & kv : kinbodies_)
      {
         const KinBody & k = kv.second;
         if (!k.enabled) continue;
         const Xform xk = xform_from_pose(body_transform(kv.first));      // (a held body is where its link is NOW)
         for (size_t v=0; v<k.tris.size()/3; v++)
         {
            double w[3];
            mat3_vec(xk.R, &k.tris[3*v], w);
            for (int r=0; r<3; r++) tris.push_back(w[r] + xk.t[r]);
         }
         for (const KinBody::B & bx : k.boxes)
         {
            Box b;
            b.world = xform_mul(xk, xform_from_pose(bx.pose));
            for (int r=0; r<3; r++) b.half[r] = bx.half[r];
            obstacles.push_back(b);
         }
      }
      // voxelize -> flood fill from cell 0 (reachable 1.0 -> 0.0, the rest becomes obstacle,
      // mod.cpp:540-548) -> signed distance field (mod.cpp:560): on the GPU for large grids
      // (or ORC_SDF_DEVICE=1), on the host otherwise; both give the same cells bit for bit
      const char * env = getenv("ORC_SDF_DEVICE");
      const bool on_device = env ? (atoi(env) != 0) : (g.ncells() >= (size_t) 1 << 18);
      if (on_device)
      {
         DeviceGuard guard(device);
         std::vector<double> bx;                       // per box: R[9] t[3] half[3]
         for (const Box & b : obstacles)
         {
            bx.insert(bx.end(), b.world.R.m, b.world.R.m + 9);
            bx.insert(bx.end(), b.world.t, b.world.t + 3);
            bx.insert(bx.end(), b.half, b.half + 3);
         }
         const Xform xg = xform_from_pose(pose_world_gsdf);
         double gx[12];
         for (int q=0; q<9; q++) gx[q] = xg.R.m[q];
         for (int q=0; q<3; q++) gx[9+q] = xg.t[q];
         hip_check(orc_sdf_build_device(g.sizes, g.lengths, gx, pose_world_gsdf.v, cube_extent, (int) obstacles.size(), bx.data(), (int)(tris.size() / 9), tris.data(),
                                        g.data.data(), stream), "sdf build");
      }
      else
      {
         voxelize_boxes(g, pose_world_gsdf, cube_extent, obstacles, tris);
         grid_flood_1_to_0(g, 0);                                                   // mod.cpp:540-548
         const size_t nc = g.ncells();
         for (size_t idx=0; idx<nc; idx++) if (g.data[idx] == 1.0) g.data[idx] = HUGE_VAL;
         Grid sdf;
         grid_bin_sdf(g, sdf);                                                      // mod.cpp:560
         g = sdf;
      }
      if (have_cache)
      {
         std::ofstream fp(cache_filename.c_str(), std::ios::binary);
         fp.write((const char *) g.data.data(), g.ncells() * sizeof(double));
      }
   }
   add_sdf(kb, g, pose_gsdf);
   return "";
}

// src/orcdchomp_mod.cpp:592-722
std::string Module::cmd_addfield_fromobsarray(const std::vector<std::string> & argv)
{
   std::string kb;
   double * obsarray = nullptr;
   int sizes[3] = {0,0,0};
   double lengths[3] = {0,0,0};
   Pose pose;
   const int argc = (int) argv.size();
   int i;
   for (i=1; i<argc; i++)
   {
      if (argv[i] == "kinbody" && i+1 < argc)
      {
         if (!kb.empty()) throw std::runtime_error("Only one kinbody can be passed!");
         kb = argv[++i];
         if (!has_body(kb)) throw std::runtime_error("Could not find kinbody with that name!");
      }
      else if (argv[i] == "obsarray" && i+1 < argc) obsarray = (double *) parse_pointer(argv[++i]);
      else if (argv[i] == "sizes" && i+1 < argc)
      {
         std::vector<std::string> t = shparse(argv[++i]);
         if (t.size() != 3) throw std::runtime_error("sizes must be length 3!");
         for (int j=0; j<3; j++) sizes[j] = std::atoi(t[j].c_str());
      }
      else if (argv[i] == "lengths" && i+1 < argc)
      {
         std::vector<double> t = parse_vector(argv[++i]);
         if (t.size() != 3) throw std::runtime_error("lengths must be length 3!");
         for (int j=0; j<3; j++) lengths[j] = t[j];
      }
      else if (argv[i] == "pose" && i+1 < argc)
      {
         std::vector<double> t = parse_vector(argv[++i]);
         if (t.size() != 7) throw std::runtime_error("pose must be length 7!");
         for (int j=0; j<7; j++) pose.v[j] = t[j];
      }
      else break;
   }
   if (i < argc) bad_arguments();
   if (kb.empty()) throw std::runtime_error("Did not pass a kinbody!");
   if (!obsarray) throw std::runtime_error("Did not pass an obsarray!");
   for (int j=0; j<3; j++) if (sizes[j] <= 0) throw std::runtime_error("Didn't pass non-zero sizes!");
   for (int j=0; j<3; j++) if (lengths[j] <= 0.0) throw std::runtime_error("Didn't pass non-zero lengths!");
   pose_normalize(pose);
   if (find_sdf(kb)) throw std::runtime_error("We already have an sdf for this kinbody!");
   Grid occ;
   for (int j=0; j<3; j++) { occ.sizes[j] = sizes[j]; occ.lengths[j] = lengths[j]; }
   // the reference takes ownership of the malloc'd array (mod.cpp:702-704); this build
   // copies it and leaves ownership with the caller (no cross-allocator free)
   occ.data.assign(obsarray, obsarray + occ.ncells());
   Grid sdf;
   bin_sdf_any(occ, sdf, stream);
   add_sdf(kb, sdf, pose);
   return "";
}

// src/orcdchomp_mod.cpp:799-847
std::string Module::cmd_removefield(const std::vector<std::string> & argv)
{
   std::string kb;
   const int argc = (int) argv.size();
   int i;
   for (i=1; i<argc; i++)
   {
      if (argv[i] == "kinbody" && i+1 < argc)
      {
         if (!kb.empty()) throw std::runtime_error("Only one kinbody can be passed!");
         kb = argv[++i];
      }
      else break;
   }
   if (i < argc) bad_arguments();
   if (kb.empty()) throw std::runtime_error("Did not pass a kinbody!");
   for (size_t k=0; k<sdfs.size(); k++)
      if (sdfs[k]->kinbody_name == kb)
      {
         sdfs.erase(sdfs.begin() + k);      // device copies live on while a batch still reads them
         return "";
      }
   throw std::runtime_error("No sdf for that kinbody!");
}

// tsr_create_parse, src/orcdchomp_mod.cpp:3068-3111: "manipindex bodyandlink" + T0w (rotation column
// by column, then translation) + Twe (the same) + Bw [6][2]; 38 fields
static bool parse_tsr(const std::string & str, TsrSpec & t)
{
   // the wire format of a TSR (tsr_create_parse, src/orcdchomp_mod.cpp:3068-3111): manipulator index, "body link"
   // word, then 36 numbers: T0w and Twe as a rotation column by column followed by the translation, Bw row by row
   std::istringstream in(str);
   int manipindex; std::string bodyandlink;
   if (!(in >> manipindex >> bodyandlink) || bodyandlink.size() > 31) return false;
   double num[36];
   for (double & v : num) if (!(in >> v)) return false;
   Pose * frames[2] = { &t.T0w, &t.Twe };
   for (int f=0; f<2; f++)
   {
      const double * block = num + 12*f;
      Mat3 R;
      for (int col=0; col<3; col++) for (int row=0; row<3; row++) R.m[3*row+col] = block[3*col+row];
      *frames[f] = pose_from_dR(block + 9, R);
   }
   for (int k=0; k<12; k++) t.Bw[k/2][k%2] = num[24+k];
   return true;
}

// src/orcdchomp_mod.cpp:1800-2688 (argument grammar 1888-2085)
std::string Module::cmd_create(const std::vector<std::string> & argv, bool batchmode)
{
   std::string rname;
   std::vector<double> adofgoal, basegoal;
   bool have_adofgoal = false, have_basegoal = false, have_starttraj = false;
   std::string starttraj;
   BatchParams p;
   unsigned int seed = 0;
   int n_runs = 1;
   std::string dat_filename;
   std::vector<int> devs; bool have_devs = false;
   std::vector<TsrSpec> con_tsrs, everyn_tsr, start_tsr;
   const double * goals_ptr = nullptr, * starts_ptr = nullptr, * basegoals_ptr = nullptr;
   const unsigned int * seeds_ptr = nullptr;
   const int argc = (int) argv.size();
   int i;
   for (i=1; i<argc; i++)
   {
      const std::string & a = argv[i];
      if (a == "robot" && i+1 < argc)
      {
         if (!rname.empty()) throw std::runtime_error("Only one robot can be passed!");
         rname = argv[++i];
         robot(rname);
      }
      else if (a == "adofgoal" && i+1 < argc)
      {
         if (have_adofgoal) throw std::runtime_error("Only one adofgoal can be passed!");
         if (have_starttraj) throw std::runtime_error("Cannot pass both adofgoal and starttraj!");
         adofgoal = parse_vector(argv[++i]); have_adofgoal = true;
      }
      else if (a == "basegoal" && i+1 < argc)
      {
         if (have_basegoal) throw std::runtime_error("Only one basegoal can be passed!");
         basegoal = parse_vector(argv[++i]);
         if (basegoal.size() != 7) throw std::runtime_error("basegoal argument must be length 7!");
         have_basegoal = true;
      }
      else if (a == "floating_base") p.floating_base = 1;
      else if (a == "lambda" && i+1 < argc) p.lambda = std::atof(argv[++i].c_str());
      else if (a == "n_points" && i+1 < argc) p.n_points = std::atoi(argv[++i].c_str());
      else if (a == "derivative" && i+1 < argc) p.derivative = std::atoi(argv[++i].c_str());
      else if (a == "use_momentum") p.use_momentum = 1;
      else if (a == "use_hmc") p.use_hmc = 1;
      else if (a == "hmc_resample_lambda" && i+1 < argc) p.hmc_resample_lambda = std::atof(argv[++i].c_str());
      else if (a == "seed" && i+1 < argc) std::sscanf(argv[++i].c_str(), "%u", &seed);
      else if (a == "epsilon" && i+1 < argc) p.epsilon = std::atof(argv[++i].c_str());
      else if (a == "epsilon_self" && i+1 < argc) p.epsilon_self = std::atof(argv[++i].c_str());
      else if (a == "obs_factor" && i+1 < argc) p.obs_factor = std::atof(argv[++i].c_str());
      else if (a == "obs_factor_self" && i+1 < argc) p.obs_factor_self = std::atof(argv[++i].c_str());
      else if (a == "no_report_cost") { /* emitted by the python layer, ignored */ }
      else if (a == "dat_filename" && i+1 < argc) dat_filename = argv[++i];
      else if (a == "starttraj" && i+1 < argc)
      {
         if (have_starttraj) throw std::runtime_error("Only one starttraj can be passed!");
         if (have_adofgoal) throw std::runtime_error("Cannot pass both adofgoal and starttraj!");
         starttraj = argv[++i]; have_starttraj = true;
      }
      else if (a == "con_tsr" && i+2 < argc)
      {
         // src/orcdchomp_mod.cpp:1930-1987: TYPE | TYPE manipee NAME | TYPE link NAME, then the TSR
         if (rname.empty()) throw std::runtime_error("You must pass robot before any con_tsrs!");
         Robot & rb = robot(rname);
         const std::vector<std::string> head = shparse(argv[++i]);
         if (head.size() != 1 && head.size() != 3) throw std::runtime_error("con_tsr first argument must be length 1 or 3!");
         if (head[0] != "all") throw std::runtime_error("con_tsr first arg must be start, end, or all!");
         TsrSpec t;
         if (head.size() != 3)
         {
            if (rb.manips.empty()) throw std::runtime_error("con_tsr manip not found!");
            t.ee_link = rb.manips[rb.active_manip].link; t.tool = rb.manips[rb.active_manip].tool;
         }
         else if (head[1] == "manipee")
         {
            size_t ui = 0;
            for (; ui<rb.manips.size(); ui++) if (rb.manips[ui].name == head[2]) break;
            if (!(ui < rb.manips.size())) throw std::runtime_error("con_tsr manip not found!");
            t.ee_link = rb.manips[ui].link; t.tool = rb.manips[ui].tool;
         }
         else if (head[1] == "link")
         {
            for (int li=0; li<rb.n_links; li++)
               if ((li < (int) rb.link_names.size() ? rb.link_names[li] : "link" + std::to_string(li)) == head[2]) t.ee_link = li;
            if (t.ee_link < 0) throw std::runtime_error("con_tsr link not found!");
         }
         else throw std::runtime_error("con_tsr first arg must be empty, manipee, or link!");
         if (!parse_tsr(argv[++i], t)) throw std::runtime_error("Cannot parse constraint TSR!");
         con_tsrs.push_back(t);
      }
      else if (a == "everyn_tsr" && i+1 < argc)
      {
         // src/orcdchomp_mod.cpp:1993-1997; applied to the active manipulator's end effector (mod.cpp:1548)
         if (rname.empty()) throw std::runtime_error("You must pass robot before any con_tsrs!");
         Robot & rb = robot(rname);
         TsrSpec t;
         if (!parse_tsr(argv[++i], t)) throw std::runtime_error("Cannot parse everyn_tsr TSR!");
         if (rb.manips.empty()) throw std::runtime_error("everyn_tsr needs an active manipulator!");
         t.ee_link = rb.manips[rb.active_manip].link; t.tool = rb.manips[rb.active_manip].tool;
         everyn_tsr.clear(); everyn_tsr.push_back(t);
      }
      else if (a == "start_tsr" && i+1 < argc)
      {
         // src/orcdchomp_mod.cpp:1988-1992: the start point becomes a variable held on this TSR by a hard
         // constraint (m = n_points-1; mod.cpp:2316-2323, 2570-2576); the active manipulator's end effector (mod.cpp:1704)
         if (rname.empty()) throw std::runtime_error("You must pass robot before any con_tsrs!");
         Robot & rb = robot(rname);
         TsrSpec t;
         if (!parse_tsr(argv[++i], t)) throw std::runtime_error("Cannot parse start_tsr TSR!");
         if (rb.manips.empty()) throw std::runtime_error("start_tsr needs an active manipulator!");
         t.ee_link = rb.manips[rb.active_manip].link; t.tool = rb.manips[rb.active_manip].tool;
         t.point = 0;
         start_tsr.clear(); start_tsr.push_back(t);
      }
      else if ((a == "start_cost" || a == "ee_force" || a == "ee_force_at" || a == "ee_torque_weights") && i+1 < argc)
         throw std::runtime_error("argument " + a + " is outside the scope of this build (SURVEY.md section 2)");
      else if (batchmode && a == "n_runs" && i+1 < argc) n_runs = std::atoi(argv[++i].c_str());
      else if (batchmode && a == "adofgoals" && i+1 < argc) goals_ptr = (const double *) parse_pointer(argv[++i]);
      else if (batchmode && a == "adofstarts" && i+1 < argc) starts_ptr = (const double *) parse_pointer(argv[++i]);
      else if (batchmode && a == "basegoals" && i+1 < argc) basegoals_ptr = (const double *) parse_pointer(argv[++i]);
      else if (batchmode && a == "seeds" && i+1 < argc) seeds_ptr = (const unsigned int *) parse_pointer(argv[++i]);
      else if (batchmode && a == "precision" && i+1 < argc) p.precision = std::atoi(argv[++i].c_str());
      else if (batchmode && a == "devices" && i+1 < argc)
      {
         for (const std::string & t : shparse(argv[++i])) devs.push_back(std::atoi(t.c_str()));
         if (devs.empty()) throw std::runtime_error("devices must name at least one device!");
         have_devs = true;
      }
      else break;
   }
   if (i < argc) bad_arguments();
   if (rname.empty()) throw std::runtime_error("Did not pass a robot!");
   if (!batchmode)
   {
      if (!have_adofgoal && !have_starttraj) throw std::runtime_error("Did not pass either adofgoal or starttraj!");
      if (p.floating_base && !have_basegoal && !have_starttraj) throw std::runtime_error("Passed floating_base with no basegoal!");
      if (!p.floating_base && have_basegoal) throw std::runtime_error("Passed basegoal with no floating_base!");
   }
   else
   {
      if (!goals_ptr) throw std::runtime_error("Did not pass either adofgoal or starttraj!");
      if (p.floating_base && !basegoals_ptr) throw std::runtime_error("Passed floating_base with no basegoal!");
      if (!p.floating_base && basegoals_ptr) throw std::runtime_error("Passed basegoal with no floating_base!");
      if (n_runs < 1) throw std::runtime_error("n_runs must be >=1!");
   }
   if (sdfs.empty()) throw std::runtime_error("No signed distance fields have yet been computed!");
   if (p.lambda < 0.01) throw std::runtime_error("lambda must be >=0.01!");
   if (p.n_points < 3) throw std::runtime_error("n_points must be >=3!");
   // a single run (the reference's `create`) gets the latency shape: eight wavefronts on the one run
   // (its trajectory is the one it has in a batch, bit for bit; its cost sums are grouped differently)
   if (!batchmode) p.workgroup_threads = 512;
   // the constraints in the reference's order of addition (src/orcdchomp_mod.cpp:2582-2612)
   if (p.floating_base && !start_tsr.empty()) throw std::runtime_error("floating_base and start_tsr together is not yet implemented!");   // mod.cpp:2100
   p.free_start = start_tsr.empty() ? 0 : 1;
   p.tsrs = start_tsr;
   p.tsrs.insert(p.tsrs.end(), everyn_tsr.begin(), everyn_tsr.end());
   p.tsrs.insert(p.tsrs.end(), con_tsrs.begin(), con_tsrs.end());
   Robot & r = robot(rname);
   // initialisation from a passed trajectory (src/orcdchomp_mod.cpp:2375-2416): sampled at
   // i * duration / (n_points-1), linear interpolation between its waypoints
   std::vector<double> sampled;
   if (!batchmode && have_starttraj)
   {
      TrajDoc doc;
      if (!parse_traj(starttraj, doc)) throw std::runtime_error("Cannot parse starttraj!");
      const int dof = doc.joint_dof, count = doc.count, c0 = p.floating_base ? 7 : 0, nn = c0 + dof;
      if (dof != (int) r.active_dofs.size()) throw std::runtime_error("size of adofgoal does not match active dofs!");
      if (p.floating_base && doc.base_off < 0) throw std::runtime_error("starttraj with floating_base needs an affine_transform group!");
      std::vector<double> tcum(count, 0.0);
      for (int k=1; k<count; k++) tcum[k] = tcum[k-1] + doc.vals[(size_t) k*doc.width + doc.dt_off];
      const double duration = tcum[count-1];
      sampled.resize((size_t) p.n_points * nn);
      int seg = 0;
      for (int k=0; k<p.n_points; k++)
      {
         // untimed documents (all deltatimes zero) are sampled uniformly over the waypoint index
         const double t = duration > 0.0 ? k * duration / (p.n_points - 1) : (double) k * (count - 1) / (p.n_points - 1);
         double u = 0.0;
         if (duration > 0.0)
         {
            while (seg < count-2 && tcum[seg+1] < t) seg++;
            const double dt = tcum[seg+1] - tcum[seg];
            u = (count > 1 && dt > 0.0) ? (t - tcum[seg]) / dt : 0.0;
         }
         else { seg = std::min((int) t, count-2); if (seg < 0) seg = 0; u = t - seg; }
         if (count == 1) { seg = 0; u = 0.0; }
         int s0 = seg, s1 = std::min(seg+1, count-1);
         // at (and past) the end a trajectory is its last waypoint (OpenRAVE's Sample), also when the last segments take no time
         if (duration > 0.0 && (k == p.n_points-1 || t >= duration)) { s0 = s1 = count-1; u = 0.0; }
         const double * r0 = &doc.vals[(size_t) s0*doc.width], * r1 = &doc.vals[(size_t) s1*doc.width];
         double * row = &sampled[(size_t) k*nn];
         if (p.floating_base)
         {
            // the base: OpenRAVE's x y z qw qx qy qz, into libcd's order, normalised (mod.cpp:2389-2399)
            double vec[7];
            for (int j=0; j<7; j++) vec[j] = r0[doc.base_off+j] + (r1[doc.base_off+j] - r0[doc.base_off+j]) * u;
            Pose bp;
            bp.v[0] = vec[0]; bp.v[1] = vec[1]; bp.v[2] = vec[2]; bp.v[3] = vec[4]; bp.v[4] = vec[5]; bp.v[5] = vec[6]; bp.v[6] = vec[3];
            pose_normalize(bp);
            for (int j=0; j<7; j++) row[j] = bp.v[j];
         }
         for (int j=0; j<dof; j++) row[c0+j] = r0[doc.joint_off+j] + (r1[doc.joint_off+j] - r0[doc.joint_off+j]) * u;
      }
      adofgoal.assign(sampled.end() - dof, sampled.end());
      if (p.floating_base) { basegoal.assign(sampled.end() - nn, sampled.end() - dof); have_basegoal = true; }
   }
   if (!batchmode && adofgoal.size() != r.active_dofs.size())
      throw std::runtime_error("size of adofgoal does not match active dofs!");
   int id;
   if (!batchmode)
   {
      id = create_batch(rname, p, 1, nullptr, adofgoal.data(), have_basegoal ? basegoal.data() : nullptr, &seed);
      if (have_starttraj) batch(id).set_traj(sampled.data());
   }
   else
      id = create_batch(rname, p, n_runs, starts_ptr, goals_ptr, basegoals_ptr, seeds_ptr, have_devs ? &devs : nullptr);
   if (!dat_filename.empty())
   {
      try { batch(id).open_dat(dat_filename); }
      catch (...) { destroy_batch(id); throw; }
   }
   std::ostringstream o;
   o << id;
   return o.str();
}

namespace {
int parse_run(const std::vector<std::string> & argv, int & i, int & run, const char * dup_msg)
{
   if (run != 0) throw std::runtime_error(dup_msg);
   int v = 0;
   if (std::sscanf(argv[++i].c_str(), "%d", &v) != 1 || v <= 0) throw std::runtime_error("Could not parse r!");
   run = v;
   return v;
}
}

// src/orcdchomp_mod.cpp:2690-2852
std::string Module::cmd_iterate(const std::vector<std::string> & argv, bool batchmode)
{
   int run = 0, n_iter = 1;
   double max_time = HUGE_VAL;
   std::string fileform;
   bool have_fileform = false;
   double * costs_ptr = nullptr; int * status_ptr = nullptr;
   const int argc = (int) argv.size();
   int i;
   for (i=1; i<argc; i++)
   {
      if (argv[i] == "run" && i+1 < argc) parse_run(argv, i, run, "Only one r can be passed!");
      else if (argv[i] == "n_iter" && i+1 < argc) n_iter = std::atoi(argv[++i].c_str());
      else if (argv[i] == "max_time" && i+1 < argc) max_time = std::atof(argv[++i].c_str());
      else if (argv[i] == "trajs_fileformstr" && i+1 < argc) { fileform = argv[++i]; have_fileform = true; }
      else if (batchmode && argv[i] == "costs" && i+1 < argc) costs_ptr = (double *) parse_pointer(argv[++i]);
      else if (batchmode && argv[i] == "status" && i+1 < argc) status_ptr = (int *) parse_pointer(argv[++i]);
      else break;
   }
   if (i < argc) bad_arguments();
   if (!run) throw std::runtime_error("you must pass a created run!");
   if (n_iter < 0) throw std::runtime_error("n_iter must be >=0!");
   Batch & b = batch(run);
   std::vector<double> costs((size_t) b.n_runs * 3, 0.0);
   std::vector<int> status(b.n_runs, 0), iters(b.n_runs, 0);
   if (have_fileform && b.params.floating_base)
      throw std::runtime_error("Error: trajs_fileformstr and floating_base combined is not yet implemented!");
   // the pattern goes to printf with (iteration) for one run as in the reference (mod.cpp:2783-2784), with
   // (iteration, run) for a batch: those integer conversions and no other.  One run may also pass a constant name (no
   // conversion: every iteration overwrites the file, which is what the reference's sprintf makes of it)
   if (have_fileform)
   {
      const int nconv = count_int_conversions(fileform);
      if (b.n_runs > 1 ? nconv != 2 : (nconv != 0 && nconv != 1)) bad_arguments();
   }
   // seconds since the call began, without the time spent writing trajectory dumps (mod.cpp:2748-2750,
   // 2781-2795: the reference stops its clock around the dump)
   auto t_last = std::chrono::steady_clock::now();
   double ticks = 0.0;
   auto clock_now = [&]() {
      const auto now = std::chrono::steady_clock::now();
      ticks += std::chrono::duration<double>(now - t_last).count();
      t_last = now;
      return ticks;
   };
   if (!have_fileform && max_time == HUGE_VAL)
   {
      b.iterate_async(n_iter);
      b.sync(costs.data(), status.data(), iters.data());
      if (b.has_dat()) b.write_dat(0, n_iter, iters.data(), 0.0, clock_now());
   }
   else
   {
      // the trajectory dump before each iteration and the time limit need the host between
      // iterations: one iteration per launch, the cost-only pass once at the end (mod.cpp:2830).
      // r->iter restarts at 0 in every iterate call while hmc_resample_iter persists (mod.cpp:2752,
      // 2755): the launches pass their position in the call on to the hmc schedule.
      std::vector<double> traj((size_t) b.n_runs * b.n_points * b.n);
      std::vector<int> st1(b.n_runs, 0);
      bool aborted = false;
      for (int it=0; it<n_iter; it++)
      {
         if (have_fileform)
         {
            clock_now();
            for (int k=0; k<b.n_runs; k++)
            {
               if (k == 0) b.gettraj(traj.data());
               char fname[1024];
               // one run: the reference's file name; a batch: the pattern takes (iteration, run)
               std::snprintf(fname, sizeof(fname), fileform.c_str(), it, k);
               std::ofstream f(fname);
               f << serialize_traj(b.robot_name, b.adofindices, &traj[(size_t) k * b.n_points * b.n], b.n_points, b.n, 0,
                                   std::vector<double>(b.n_points, 0.0));
            }
            t_last = std::chrono::steady_clock::now();        // the dump is off the clock
         }
         const double t_begin = ticks;
         const std::vector<int> before = (it > 0) ? iters : std::vector<int>(b.n_runs, 0);
         b.iterate_async(1, it, false, it > 0);                // (runs that left their limits earlier in the call stay out)
         b.sync(costs.data(), st1.data(), iters.data());       // iterations made accumulate over the launches of the call
         const double t_end = clock_now();
         if (b.has_dat())
         {
            std::vector<int> made(b.n_runs);
            for (int k=0; k<b.n_runs; k++) made[k] = iters[k] - before[k];
            b.write_dat(it, 1, made.data(), t_begin, t_end);
         }
         for (int k=0; k<b.n_runs; k++) if (st1[k] != 0) { status[k] = st1[k]; aborted = true; }
         // a single run stops where the reference throws; a batch goes on for its other runs
         if (aborted && b.n_runs == 1) break;
         if (t_end > max_time) break;
      }
      if (!(aborted && b.n_runs == 1))
      {
         b.iterate_async(0, 0, true, n_iter > 0);              // cd_chomp_iterate(c, 0, ...) (mod.cpp:2830)
         b.sync(costs.data(), st1.data(), nullptr);
      }
   }
   if (costs_ptr) std::memcpy(costs_ptr, costs.data(), costs.size() * sizeof(double));
   if (status_ptr) std::memcpy(status_ptr, status.data(), status.size() * sizeof(int));
   if (!batchmode)
      for (int k=0; k<b.n_runs; k++)
         if (status[k] == -1) throw std::runtime_error("Resulting trajectory is outside of joint limits!");
   std::ostringstream o;
   o << costs[0];                                                   // sout << cost_total (mod.cpp:2849)
   return o.str();
}

// Samples of a retimed trajectory every 0.04 rad of C-space distance, the grid of the reference's
// re-check (src/orcdchomp_mod.cpp:2958-3006): for every sample the segment it lies on, the position
// on the segment and its time.
static void plan_collision_samples(const double * traj, int n_points, int n, int col0, const std::vector<double> & dtm,
   std::vector<int> & seg_out, std::vector<double> & u_out, std::vector<double> & time_out)
{
   double total_dist = 0.0, duration = 0.0;
   for (int i=0; i+1<n_points; i++)
   {
      double d2 = 0.0;
      for (int j=col0; j<n; j++) { const double d = traj[(size_t) i*n+j] - traj[(size_t)(i+1)*n+j]; d2 += d*d; }
      total_dist += std::sqrt(d2);
      duration += dtm[i+1];
   }
   const double step_time = total_dist > 0.0 ? duration * 0.04 / total_dist : duration + 1.0;
   int seg = 0; double tseg0 = 0.0;
   for (double time=0.0; time<duration; time+=step_time)
   {
      while (seg < n_points-2 && tseg0 + dtm[seg+1] < time) { tseg0 += dtm[seg+1]; seg++; }
      const double u = dtm[seg+1] > 0.0 ? (time - tseg0) / dtm[seg+1] : 0.0;
      seg_out.push_back(seg); u_out.push_back(u); time_out.push_back(time);
   }
}

void Module::batch_collision_verdict(int id, int * collides, double * time, int * sphere, int * field, double * depth, bool self_check)
{
   Batch & b = batch(id);
   const int col0 = b.params.floating_base ? 7 : 0;
   Robot rob = robot(b.robot_name);
   rob.spheres = b.run_spheres;         // the robot's and those of the bodies it held at create (mod.cpp:2992-2996)
   std::vector<double> vmax;
   for (int a : b.adofindices) vmax.push_back(a < (int) rob.limit_vel.size() ? rob.limit_vel[a] : 1.0);
   std::vector<double> traj((size_t) b.n_runs * b.n_points * b.n);
   b.gettraj(traj.data());
   std::vector<int> offs(b.n_runs + 1, 0), seg;
   std::vector<double> u, times;
   for (int k=0; k<b.n_runs; k++)
   {
      const double * tk = &traj[(size_t) k * b.n_points * b.n];
      const std::vector<double> dtm = retime_linear(tk, b.n_points, b.n, col0, vmax);
      plan_collision_samples(tk, b.n_points, b.n, col0, dtm, seg, u, times);
      if (seg.size() >= ((size_t) 1 << 31) - 1) throw std::runtime_error("trajectory too long for the batched collision verdict!");      // (the running total is an int on both sides)
      offs[k+1] = (int) seg.size();
      if (offs[k+1] - offs[k] >= (1 << 30)) throw std::runtime_error("trajectory too long for the batched collision verdict!");
   }
   // the pairs of the self-collision leg (`|| CheckSelfCollision`, mod.cpp:2998-2999): spheres on links that may
   // collide, in XML order; an end is a slot of the device's position row or an inactive sphere's world position
   std::vector<int> pairs; std::vector<double> rsum, inact_pos;
   if ((int) rob.spheres.size() > 128) throw std::runtime_error("too many spheres for the batched collision verdict!");
   if (self_check && rob.self_check)
   {
      const std::vector<unsigned char> & excl = b.run_self_excl;      // sphere by sphere, taken at create (held bodies: Robot::run_self_pairs_excluded)
      const int ns = (int) rob.spheres.size();
      std::vector<int> end_of(ns, 0);
      std::vector<Xform> frames;
      rob.fk(rob.transform, rob.dof_values, frames);
      for (int si=0; si<ns; si++)
      {
         int slot = -1;
         for (size_t q=0; q<b.slot_xml.size(); q++) if (b.slot_xml[q] == si) slot = (int) q;
         if (slot >= 0) { end_of[si] = slot; continue; }
         end_of[si] = -1 - (int)(inact_pos.size() / 3);
         double r[3];
         mat3_vec(frames[rob.spheres[si].link].R, rob.spheres[si].pos, r);
         for (int q=0; q<3; q++) inact_pos.push_back(r[q] + frames[rob.spheres[si].link].t[q]);
      }
      for (int a=0; a<ns; a++)
         for (int c=a+1; c<ns; c++)
         {
            if (excl[(size_t) a * ns + c]) continue;
            pairs.push_back(end_of[a]); pairs.push_back(end_of[c]); pairs.push_back(a); pairs.push_back(c);
            rsum.push_back(rob.spheres[a].radius + rob.spheres[c].radius);
         }
   }
   std::vector<unsigned long long> key(b.n_runs); std::vector<double> dep(b.n_runs);
   b.collision_verdict(offs, seg, u, pairs, rsum, inact_pos, key.data(), dep.data());
   if (getenv("ORC_DEBUG_VERDICT"))
      for (int k=0; k<b.n_runs; k++) fprintf(stderr, "verdict run %d key %016llx samples %d\n", k, key[k], offs[k+1] - offs[k]);
   for (int k=0; k<b.n_runs; k++)
   {
      // key: sample << 32 | pair bit << 31 | XML sphere << 16 | field (or, for a pair, the other sphere)
      const bool hit = key[k] != ORC_VERDICT_NONE;
      const bool self = hit && ((key[k] >> 31) & 1ull);
      if (collides) collides[k] = hit ? 1 : 0;
      if (time) time[k] = hit ? times[(size_t) offs[k] + (size_t)(key[k] >> 32)] : -1.0;
      if (sphere) sphere[k] = hit ? (int)((key[k] >> 16) & 0x7fffull) : -1;
      if (field) field[k] = hit ? (self ? -2 - (int)(key[k] & 0xffffull) : (int)(key[k] & 0xffffull)) : -1;      // a pair: -2 - the other sphere
      if (depth) depth[k] = hit ? dep[k] : 0.0;
   }
}

// src/orcdchomp_mod.cpp:2854-3011
std::string Module::cmd_gettraj(const std::vector<std::string> & argv, bool batchmode)
{
   int run = 0;
   bool no_collision_check = false, no_collision_exception = false, no_collision_details = false, no_self_check = false;
   double * out_ptr = nullptr;
   int * verdict_ptr = nullptr;
   const int argc = (int) argv.size();
   int i;
   for (i=1; i<argc; i++)
   {
      if (argv[i] == "run" && i+1 < argc) parse_run(argv, i, run, "Only one r can be passed!");
      else if (argv[i] == "no_collision_check") no_collision_check = true;
      else if (argv[i] == "no_collision_exception") no_collision_exception = true;
      else if (argv[i] == "no_collision_details") no_collision_details = true;
      else if (argv[i] == "no_self_collision_check") no_self_check = true;      // additive: the field leg of the re-check alone
      else if (batchmode && argv[i] == "out" && i+1 < argc) out_ptr = (double *) parse_pointer(argv[++i]);
      else if (batchmode && argv[i] == "verdict" && i+1 < argc) verdict_ptr = (int *) parse_pointer(argv[++i]);
      else break;
   }
   if (i < argc) bad_arguments();
   if (!run) throw std::runtime_error("you must pass a created run!");
   Batch & b = batch(run);
   std::vector<double> traj((size_t) b.n_runs * b.n_points * b.n);
   b.gettraj(traj.data());
   if (batchmode)
   {
      if (!out_ptr) throw std::runtime_error("gettrajbatch needs out %p!");
      std::memcpy(out_ptr, traj.data(), traj.size() * sizeof(double));
      if (verdict_ptr) batch_collision_verdict(run, verdict_ptr, nullptr, nullptr, nullptr, nullptr, !no_self_check);
      return "";
   }
   const int col0 = b.params.floating_base ? 7 : 0;
   Robot rob = robot(b.robot_name);
   rob.spheres = b.run_spheres;
   std::vector<double> vmax;
   for (int a : b.adofindices) vmax.push_back(a < (int) rob.limit_vel.size() ? rob.limit_vel[a] : 1.0);
   // timing (mod.cpp:2905-2911)
   const std::vector<double> dtm = retime_linear(traj.data(), b.n_points, b.n, col0, vmax);
   if (!no_collision_check)
   {
      // The reference samples the timed trajectory every 0.04 rad of C-space distance and asks
      // OpenRAVE for environment/self collisions (mod.cpp:2958-3006).  Here the verdict comes from
      // the model the optimizer itself uses: a configuration collides when an active sphere
      // penetrates a signed distance field (interpolated field value below the sphere radius).
      double total_dist = 0.0, duration = 0.0;
      for (int i=0; i+1<b.n_points; i++)
      {
         double d2 = 0.0;
         for (int j=col0; j<b.n; j++) { const double d = traj[(size_t) i*b.n+j] - traj[(size_t)(i+1)*b.n+j]; d2 += d*d; }
         total_dist += std::sqrt(d2);
         duration += dtm[i+1];
      }
      const double step_time = total_dist > 0.0 ? duration * 0.04 / total_dist : duration + 1.0;
      std::vector<Xform> frames;
      std::vector<double> q = rob.dof_values;
      const std::vector<unsigned char> & self_excl = b.run_self_excl;
      const size_t n_run_spheres = rob.spheres.size();
      bool collides = false;
      std::ostringstream details;
      int seg = 0; double tseg0 = 0.0;
      for (double time=0.0; time<duration && !(collides && !no_collision_exception); time+=step_time)
      {
         while (seg < b.n_points-2 && tseg0 + dtm[seg+1] < time) { tseg0 += dtm[seg+1]; seg++; }
         const double u = dtm[seg+1] > 0.0 ? (time - tseg0) / dtm[seg+1] : 0.0;
         Pose base = rob.transform;
         if (col0)
         {
            for (int j=0; j<7; j++) base.v[j] = traj[(size_t) seg*b.n+j] + (traj[(size_t)(seg+1)*b.n+j] - traj[(size_t) seg*b.n+j]) * u;
            pose_normalize(base);
         }
         for (size_t a=0; a<b.adofindices.size(); a++)
         {
            const double a0 = traj[(size_t) seg*b.n+col0+a], a1 = traj[(size_t)(seg+1)*b.n+col0+a];
            q[b.adofindices[a]] = a0 + (a1 - a0) * u;
         }
         rob.fk(base, q, frames);
         for (size_t si=0; si<rob.spheres.size() && !collides; si++)
         {
            const Robot::Sphere & sp = rob.spheres[si];
            bool active = col0 != 0;
            for (size_t a=0; a<b.adofindices.size() && !active; a++) active = rob.does_affect(b.adofindices[a], sp.link);
            if (!active) continue;
            double pw[3];
            mat3_vec(frames[sp.link].R, sp.pos, pw);
            for (int k=0; k<3; k++) pw[k] += frames[sp.link].t[k];
            for (auto & f : sdfs)
            {
               const Pose pose_world_gsdf = pose_compose(body_transform(f->kinbody_name), f->pose);
               double pg[3], val;
               pose_apply(pose_invert(pose_world_gsdf), pw, pg);
               if (grid_interp(f->grid, pg, &val)) continue;
               if (val - sp.radius < 0.0)
               {
                  collides = true;
                  details << "Collision at t=" << time << ": sphere " << si << " of " << b.robot_name
                          << " is " << (sp.radius - val) << " m inside the field of " << f->kinbody_name << "\n";
                  break;
               }
            }
         }
         // ... || CheckSelfCollision (mod.cpp:2998-2999): two spheres on links that may collide overlap
         for (size_t a=0; a<rob.spheres.size() && !collides && !no_self_check && rob.self_check; a++)
            for (size_t c=a+1; c<rob.spheres.size(); c++)
            {
               const Robot::Sphere & sa = rob.spheres[a], & sc = rob.spheres[c];
               if (self_excl[a * n_run_spheres + c]) continue;
               double pa[3], pc[3], d2 = 0.0;
               mat3_vec(frames[sa.link].R, sa.pos, pa);
               mat3_vec(frames[sc.link].R, sc.pos, pc);
               for (int k=0; k<3; k++) { const double d = (pa[k] + frames[sa.link].t[k]) - (pc[k] + frames[sc.link].t[k]); d2 += d*d; }
               const double dist = std::sqrt(d2), rs = sa.radius + sc.radius;
               if (dist - rs < 0.0)
               {
                  collides = true;
                  details << "Collision at t=" << time << ": spheres " << a << " and " << c << " of " << b.robot_name
                          << " overlap by " << (rs - dist) << " m\n";
                  break;
               }
            }
      }
      last_collision_details = no_collision_details ? std::string() : details.str();
      if (collides && !no_collision_exception) throw std::runtime_error("Resulting trajectory is in collision!");
   }
   // active dof columns only (mod.cpp:2899-2903)
   return serialize_traj(b.robot_name, b.adofindices, traj.data(), b.n_points, b.n, col0, dtm);
}

// src/orcdchomp_mod.cpp:3013-3037
std::string Module::cmd_destroy(const std::vector<std::string> & argv)
{
   int run = 0;
   const int argc = (int) argv.size();
   int i;
   for (i=1; i<argc; i++)
   {
      if (argv[i] == "run" && i+1 < argc) parse_run(argv, i, run, "Only one run can be passed!");
      else break;
   }
   if (i < argc) bad_arguments();
   if (!run) throw std::runtime_error("you must pass a created run!");
   destroy_batch(run);
   return "";
}

} // namespace orc
