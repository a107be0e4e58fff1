// capi.cpp -- the extern "C" boundary declared in include/orcdchomp_amd.h.
#include "../../include/orcdchomp_amd.h"
#include "module.h"
#include <cstring>
#include <new>
#include <stdexcept>

struct orc_module
{
   orc::Module * impl;
   std::string last_error;
};

namespace {
std::string g_new_error;

template <typename F>
int guarded(orc_module * mod, F f)
{
   if (!mod) return 2;
   // every entry point asserts the module's (first) device for the calling thread: modules on
   // different GPUs may share a process, and callers may come from any thread
   try { orc::DeviceGuard guard(mod->impl->device); f(); mod->last_error.clear(); return 0; }
   catch (const std::exception & e) { mod->last_error = e.what(); return 1; }
   catch (...) { mod->last_error = "unknown error"; return 1; }
}

// a C caller's mistakes come back as error codes, never as a fault inside the library
void need(const void * p, const char * what)
{
   if (!p) throw std::runtime_error(std::string("null argument: ") + what);
}
const char * str(const char * s, const char * what) { need(s, what); return s; }
}

extern "C" {

orc_module * orc_module_new(int device)
{
   try
   {
      orc_module * m = new orc_module;
      m->impl = new orc::Module(device);
      return m;
   }
   catch (const std::exception & e) { g_new_error = e.what(); return nullptr; }
}

orc_module * orc_module_new_multi(const int * devices, int n_devices)
{
   try
   {
      if (!devices || n_devices < 1) throw std::runtime_error("orcdchomp_amd: empty device list");
      orc_module * m = new orc_module;
      m->impl = new orc::Module(std::vector<int>(devices, devices + n_devices));
      return m;
   }
   catch (const std::exception & e) { g_new_error = e.what(); return nullptr; }
}

void orc_module_free(orc_module * mod)
{
   if (!mod) return;
   delete mod->impl;
   delete mod;
}

const char * orc_last_error(const orc_module * mod)
{
   if (!mod) return g_new_error.c_str();
   return mod->last_error.c_str();
}

int orc_set_stream(orc_module * mod, void * hip_stream)
{
   return guarded(mod, [&] { mod->impl->stream = (hipStream_t) hip_stream; });
}

int orc_set_num_streams(orc_module * mod, int n)
{
   return guarded(mod, [&] { mod->impl->set_num_streams(n < 0 ? 0 : n); });
}

int orc_send_command(orc_module * mod, const char * cmd, char * out, size_t out_cap)
{
   return guarded(mod, [&] {
      mod->impl->last_reply = mod->impl->send_command(cmd ? cmd : "");
      if (out && out_cap)
      {
         const size_t k = std::min(out_cap - 1, mod->impl->last_reply.size());
         std::memcpy(out, mod->impl->last_reply.data(), k);
         out[k] = 0;
      }
   });
}

size_t orc_last_reply_size(const orc_module * mod) { return mod ? mod->impl->last_reply.size() : 0; }

int orc_last_reply(const orc_module * mod, char * out, size_t out_cap)
{
   if (!mod || !out || !out_cap) return 2;
   const size_t k = std::min(out_cap - 1, mod->impl->last_reply.size());
   std::memcpy(out, mod->impl->last_reply.data(), k);
   out[k] = 0;
   return 0;
}

int orc_env_add_robot(orc_module * mod, const char * name, const orc_robot_desc * d)
{
   return guarded(mod, [&] {
      need(name, "name"); need(d, "robot description");
      if (d->n_links < 1 || d->n_dof < 0 || d->n_spheres < 0) throw std::runtime_error("bad robot description: counts!");
      need(d->parent, "parent"); need(d->pose_parent_joint, "pose_parent_joint"); need(d->joint_type, "joint_type");
      need(d->axis, "axis"); need(d->dof_index, "dof_index");
      if (d->n_dof) { need(d->limit_lower, "limit_lower"); need(d->limit_upper, "limit_upper"); }
      if (d->n_spheres) { need(d->sphere_link, "sphere_link"); need(d->sphere_pos, "sphere_pos"); need(d->sphere_radius, "sphere_radius"); }
      for (int i=0; i<d->n_links; i++)
      {
         const int jt = d->joint_type[i], dof = d->dof_index[i];
         if (jt < 0 || jt > 2) throw std::runtime_error("bad robot description: joint type (0 fixed, 1 revolute, 2 prismatic)!");
         if (jt == 0 ? dof != -1 : (dof < 0 || dof >= d->n_dof)) throw std::runtime_error("bad robot description: dof index of a joint!");
         const double * a = d->axis + 3*i;
         const double a2 = a[0]*a[0] + a[1]*a[1] + a[2]*a[2];
         if (jt != 0 && !(a2 > 0.0 && a2 < 1e300)) throw std::runtime_error("bad robot description: joint axis!");
      }
      orc::Robot r;
      r.name = name;
      r.n_links = d->n_links;
      r.parent.assign(d->parent, d->parent + d->n_links);
      for (int i=0; i<d->n_links; i++)
      {
         if (r.parent[i] >= i) throw std::runtime_error("links must be in topological order!");
         r.pose_parent_joint.push_back(orc::Pose(d->pose_parent_joint + 7*i));
      }
      r.joint_type.assign(d->joint_type, d->joint_type + d->n_links);
      r.axis.assign(d->axis, d->axis + 3*d->n_links);
      r.dof_index.assign(d->dof_index, d->dof_index + d->n_links);
      r.n_dof = d->n_dof;
      r.limit_lower.assign(d->limit_lower, d->limit_lower + d->n_dof);
      r.limit_upper.assign(d->limit_upper, d->limit_upper + d->n_dof);
      for (int s=0; s<d->n_spheres; s++)
      {
         orc::Robot::Sphere sp;
         sp.link = d->sphere_link[s];
         if (sp.link < 0 || sp.link >= d->n_links) throw std::runtime_error("link in <orcdchomp> does not exist.");
         for (int k=0; k<3; k++) sp.pos[k] = d->sphere_pos[3*s+k];
         sp.radius = d->sphere_radius[s];
         if (!(sp.radius > 0.0 && sp.radius < 1e300)) throw std::runtime_error("bad robot description: sphere radius!");
         r.spheres.push_back(sp);
      }
      r.dof_values.assign(d->n_dof, 0.0);
      for (int i=0; i<d->n_dof; i++) r.active_dofs.push_back(i);
      mod->impl->add_robot(r);
   });
}

int orc_robot_set_transform(orc_module * mod, const char * name, const double pose[7])
{
   return guarded(mod, [&] { need(pose, "pose"); mod->impl->robot(str(name, "name")).transform = orc::Pose(pose); });
}

int orc_robot_set_dof_values(orc_module * mod, const char * name, const double * values, int n)
{
   return guarded(mod, [&] {
      orc::Robot & r = mod->impl->robot(str(name, "name"));
      if (n != r.n_dof) throw std::runtime_error("wrong number of dof values!");
      if (n) need(values, "values");
      r.dof_values.assign(values, values + n);
   });
}

int orc_robot_set_active_dofs(orc_module * mod, const char * name, const int * indices, int n)
{
   return guarded(mod, [&] {
      orc::Robot & r = mod->impl->robot(str(name, "name"));
      if (n < 0) throw std::runtime_error("bad dof index!");
      if (n) need(indices, "indices");
      for (int i=0; i<n; i++) if (indices[i] < 0 || indices[i] >= r.n_dof) throw std::runtime_error("bad dof index!");
      r.active_dofs.assign(indices, indices + n);
   });
}

int orc_robot_set_velocity_limits(orc_module * mod, const char * name, const double * limits, int n)
{
   return guarded(mod, [&] {
      orc::Robot & r = mod->impl->robot(str(name, "name"));
      if (n != r.n_dof) throw std::runtime_error("wrong number of velocity limits!");
      if (n) need(limits, "limits");
      r.limit_vel.assign(limits, limits + n);
   });
}

int orc_set_workgroup_threads(orc_module * mod, int threads)
{
   return guarded(mod, [&] {
      if (threads != 0 && threads != 128 && threads != 192 && threads != 256 && threads != 512)
         throw std::runtime_error("workgroup threads must be 0 (default), 128, 192, 256 or 512!");
      mod->impl->workgroup_threads = threads;
   });
}

int orc_set_workgroups_per_cu(orc_module * mod, int workgroups)
{
   return guarded(mod, [&] {
      if (workgroups != 0 && workgroups != 3 && workgroups != 4) throw std::runtime_error("workgroups per CU must be 0 (the planner's choice), 3 or 4!");
      mod->impl->workgroups_per_cu = workgroups;
   });
}

int orc_robot_set_link_names(orc_module * mod, const char * name, const char * const * names, int n)
{
   return guarded(mod, [&] {
      orc::Robot & r = mod->impl->robot(str(name, "name"));
      if (n != r.n_links) throw std::runtime_error("wrong number of link names!");
      need(names, "names");
      for (int i=0; i<n; i++) need(names[i], "a link name");
      r.link_names.assign(names, names + n);
   });
}

int orc_robot_add_manipulator(orc_module * mod, const char * name, const char * manip, int ee_link, const double tool_pose[7])
{
   return guarded(mod, [&] {
      orc::Robot & r = mod->impl->robot(str(name, "name"));
      if (ee_link < 0 || ee_link >= r.n_links) throw std::runtime_error("manipulator end-effector link out of range!");
      orc::Robot::Manip m;
      m.name = str(manip, "manipulator name"); m.link = ee_link;
      if (tool_pose) m.tool = orc::Pose(tool_pose);
      r.manips.push_back(m);
   });
}

int orc_robot_set_adjacent_links(orc_module * mod, const char * name, const int * link_pairs, int n_pairs)
{
   return guarded(mod, [&] {
      orc::Robot & r = mod->impl->robot(str(name, "name"));
      if (n_pairs < 0 || (n_pairs > 0 && !link_pairs)) throw std::runtime_error("bad adjacent link list!");
      r.adjacent.clear();
      for (int k=0; k<n_pairs; k++)
      {
         const int a = link_pairs[2*k], b = link_pairs[2*k+1];
         if (a < 0 || a >= r.n_links || b < 0 || b >= r.n_links) throw std::runtime_error("adjacent link out of range!");
         r.adjacent.push_back(std::make_pair(a, b));
      }
   });
}

int orc_robot_set_self_check(orc_module * mod, const char * name, int enabled)
{
   return guarded(mod, [&] { mod->impl->robot(str(name, "name")).self_check = enabled != 0; });
}

int orc_robot_set_active_manipulator(orc_module * mod, const char * name, const char * manip)
{
   return guarded(mod, [&] {
      orc::Robot & r = mod->impl->robot(str(name, "name"));
      for (size_t k=0; k<r.manips.size(); k++)
         if (r.manips[k].name == str(manip, "manipulator name")) { r.active_manip = (int) k; return; }
      throw std::runtime_error("manipulator not found!");
   });
}

int orc_env_add_kinbody_boxes(orc_module * mod, const char * name, int n_boxes, const double * box_poses, const double * half_extents)
{
   return guarded(mod, [&] {
      need(name, "name");
      if (n_boxes < 0) throw std::runtime_error("bad number of boxes!");
      if (n_boxes) { need(box_poses, "box_poses"); need(half_extents, "half_extents"); }
      orc::KinBody k;
      k.name = name;
      for (int i=0; i<n_boxes; i++)
      {
         orc::KinBody::B b;
         b.pose = orc::Pose(box_poses + 7*i);
         for (int q=0; q<3; q++) b.half[q] = half_extents[3*i+q];
         k.boxes.push_back(b);
      }
      mod->impl->add_kinbody(k);
   });
}

int orc_env_add_kinbody_trimesh(orc_module * mod, const char * name, int n_tri, const double * vertices)
{
   return guarded(mod, [&] {
      if (n_tri < 1) throw std::runtime_error("a mesh needs at least one triangle!");
      need(vertices, "vertices");
      if (n_tri > (1 << 26)) throw std::runtime_error("too many triangles in one mesh (at most 2^26)!");
      for (size_t i=0; i<9*(size_t) n_tri; i++) if (!std::isfinite(vertices[i])) throw std::runtime_error("mesh vertices must be finite numbers!");
      const std::string nm = str(name, "name");
      if (mod->impl->has_body(nm))
      {
         // a kinbody of boxes gets its mesh geometry added (a body may have both kinds); a robot of that name is an error.
         // A body whose distance field exists already keeps the field of its OLD geometry (computedistancefield refuses a second
         // one: "We already have an sdf for this kinbody!"), so new geometry for it is an error here, not a stale field later
         if (mod->impl->find_sdf(nm)) throw std::runtime_error("that kinbody has a distance field already: removefield first!");
         orc::KinBody & k = mod->impl->kinbody(nm);
         k.tris.insert(k.tris.end(), vertices, vertices + 9*(size_t) n_tri);
         return;
      }
      orc::KinBody k;
      k.name = nm;
      k.tris.assign(vertices, vertices + 9*(size_t) n_tri);
      mod->impl->add_kinbody(k);
   });
}

int orc_kinbody_set_transform(orc_module * mod, const char * name, const double pose[7])
{
   return guarded(mod, [&] { need(pose, "pose"); mod->impl->set_kinbody_transform(str(name, "name"), orc::Pose(pose)); });
}

int orc_kinbody_set_spheres(orc_module * mod, const char * name, int n_spheres, const double * sphere_pos, const double * sphere_radius)
{
   return guarded(mod, [&] {
      orc::KinBody & k = mod->impl->kinbody(str(name, "name"));
      if (n_spheres < 0) throw std::runtime_error("bad number of spheres!");
      if (n_spheres) { need(sphere_pos, "sphere_pos"); need(sphere_radius, "sphere_radius"); }
      k.spheres.clear();
      for (int i=0; i<n_spheres; i++)
      {
         orc::Robot::Sphere sp;
         sp.link = 0; sp.radius = sphere_radius[i];
         for (int q=0; q<3; q++) sp.pos[q] = sphere_pos[3*i+q];
         k.spheres.push_back(sp);
      }
      mod->impl->refresh_grab_contacts(k.name);      // (a held body's contacts were taken with its old spheres)
   });
}

int orc_robot_grab(orc_module * mod, const char * robot, const char * kinbody, int link)
{
   return guarded(mod, [&] { mod->impl->grab(str(robot, "robot"), str(kinbody, "kinbody"), link); });
}

int orc_robot_release(orc_module * mod, const char * robot, const char * kinbody)
{
   return guarded(mod, [&] { mod->impl->release(str(robot, "robot"), str(kinbody, "kinbody")); });
}

int orc_robot_release_all(orc_module * mod, const char * robot)
{
   return guarded(mod, [&] { mod->impl->release_all(str(robot, "robot")); });
}

int orc_body_get_transform(orc_module * mod, const char * name, double pose_out[7])
{
   return guarded(mod, [&] {
      need(pose_out, "pose_out");
      if (!mod->impl->has_body(str(name, "name"))) throw std::runtime_error("Could not find kinbody with that name!");
      const orc::Pose p = mod->impl->body_transform(name);
      for (int i=0; i<7; i++) pose_out[i] = p.v[i];
   });
}

int orc_kinbody_enable(orc_module * mod, const char * name, int enabled)
{
   return guarded(mod, [&] { mod->impl->kinbody(str(name, "name")).enabled = enabled != 0; });
}

int orc_scene_add_sdf(orc_module * mod, const char * kinbody, const int sizes[3], const double lengths[3],
   const double pose[7], const double * data)
{
   return guarded(mod, [&] {
      need(kinbody, "kinbody"); need(sizes, "sizes"); need(lengths, "lengths"); need(pose, "pose"); need(data, "data");
      if (!mod->impl->has_body(kinbody)) throw std::runtime_error("Could not find kinbody with that name!");
      for (int i=0; i<3; i++) if (sizes[i] < 2 || !(lengths[i] > 0.0)) throw std::runtime_error("sdf grids need at least 2 cells per dimension and positive lengths!");
      orc::Grid g;
      for (int i=0; i<3; i++) { g.sizes[i] = sizes[i]; g.lengths[i] = lengths[i]; }
      g.data.assign(data, data + g.ncells());
      mod->impl->add_sdf(kinbody, g, orc::Pose(pose));
   });
}

int orc_scene_get_sdf(orc_module * mod, const char * kinbody, int sizes[3], double lengths[3], double pose[7],
   double * data, size_t data_cap)
{
   return guarded(mod, [&] {
      need(kinbody, "kinbody"); need(sizes, "sizes"); need(lengths, "lengths"); need(pose, "pose");
      orc::Sdf * s = mod->impl->find_sdf(kinbody);
      if (!s) throw std::runtime_error("No sdf for that kinbody!");
      for (int i=0; i<3; i++) { sizes[i] = s->grid.sizes[i]; lengths[i] = s->grid.lengths[i]; }
      for (int i=0; i<7; i++) pose[i] = s->pose.v[i];
      if (data)
      {
         if (data_cap < s->grid.ncells()) throw std::runtime_error("buffer too small!");
         std::memcpy(data, s->grid.data.data(), s->grid.ncells() * sizeof(double));
      }
   });
}

void orc_batch_params_default(orc_batch_params * p)
{
   if (!p) return;
   orc::BatchParams d;
   p->n_points = d.n_points; p->floating_base = d.floating_base; p->lambda = d.lambda;
   p->derivative = d.derivative; p->use_momentum = d.use_momentum; p->use_hmc = d.use_hmc;
   p->hmc_resample_lambda = d.hmc_resample_lambda; p->epsilon = d.epsilon; p->epsilon_self = d.epsilon_self;
   p->obs_factor = d.obs_factor; p->obs_factor_self = d.obs_factor_self; p->precision = d.precision;
}

int orc_batch_create(orc_module * mod, const char * robot, const orc_batch_params * p, int n_runs,
   const double * starts, const double * goals, const double * basegoals, const unsigned int * seeds, int * batch_id)
{
   return guarded(mod, [&] {
      need(robot, "robot"); need(p, "params"); need(batch_id, "batch_id");
      orc::BatchParams q;
      q.n_points = p->n_points; q.floating_base = p->floating_base; q.lambda = p->lambda;
      q.derivative = p->derivative; q.use_momentum = p->use_momentum; q.use_hmc = p->use_hmc;
      q.hmc_resample_lambda = p->hmc_resample_lambda; q.epsilon = p->epsilon; q.epsilon_self = p->epsilon_self;
      q.obs_factor = p->obs_factor; q.obs_factor_self = p->obs_factor_self; q.precision = p->precision;
      // the same validation the create command performs (mod.cpp:2091-2097)
      if (!goals) throw std::runtime_error("Did not pass either adofgoal or starttraj!");
      if (q.floating_base && !basegoals) throw std::runtime_error("Passed floating_base with no basegoal!");
      if (!q.floating_base && basegoals) throw std::runtime_error("Passed basegoal with no floating_base!");
      if (mod->impl->sdfs.empty()) throw std::runtime_error("No signed distance fields have yet been computed!");
      if (q.lambda < 0.01) throw std::runtime_error("lambda must be >=0.01!");
      if (q.n_points < 3) throw std::runtime_error("n_points must be >=3!");
      if (n_runs < 1) throw std::runtime_error("n_runs must be >=1!");
      *batch_id = mod->impl->create_batch(robot, q, n_runs, starts, goals, basegoals, seeds);
   });
}

int orc_batch_iterate(orc_module * mod, int id, int n_iter, double * costs_out, int * status_out)
{
   return guarded(mod, [&] {
      orc::Batch & b = mod->impl->batch(id);
      b.iterate_async(n_iter);
      b.sync(costs_out, status_out);
   });
}

int orc_batch_iterate_async(orc_module * mod, int id, int n_iter)
{
   return guarded(mod, [&] { mod->impl->batch(id).iterate_async(n_iter); });
}

int orc_batch_sync(orc_module * mod, int id, double * costs_out, int * status_out)
{
   return guarded(mod, [&] { mod->impl->batch(id).sync(costs_out, status_out); });
}

int orc_batch_iterations_done(orc_module * mod, int id, int * iters_out)
{
   return guarded(mod, [&] { mod->impl->batch(id).sync(nullptr, nullptr, iters_out); });
}

int orc_batch_get_trace(orc_module * mod, int id, double * out, size_t cap)
{
   return guarded(mod, [&] {
      orc::Batch & b = mod->impl->batch(id);
      need(out, "out");
      if (cap < (size_t) b.n_runs * b.last_n_iter * 3) throw std::runtime_error("buffer too small!");
      b.get_trace(out);
   });
}

int orc_batch_set_noise(orc_module * mod, int id, const double * noise, int n_blocks)
{
   return guarded(mod, [&] {
      if (n_blocks < 0 || (n_blocks > 0 && !noise)) throw std::runtime_error("bad noise blocks!");
      mod->impl->batch(id).set_noise(noise, n_blocks);
   });
}

int orc_batch_gettraj(orc_module * mod, int id, double * out, size_t cap)
{
   return guarded(mod, [&] {
      orc::Batch & b = mod->impl->batch(id);
      need(out, "out");
      if (cap < (size_t) b.n_runs * b.n_points * b.n) throw std::runtime_error("buffer too small!");
      b.gettraj(out);
   });
}

int orc_batch_collision_verdict(orc_module * mod, int id, int * collides_out, double * time_out, int * sphere_out,
   int * field_out, double * depth_out)
{
   return guarded(mod, [&] {
      need(collides_out, "collides_out");
      mod->impl->batch_collision_verdict(id, collides_out, time_out, sphere_out, field_out, depth_out);
   });
}

int orc_batch_get_state(orc_module * mod, int id, const char * which, double * out, size_t cap)
{
   return guarded(mod, [&] {
      orc::Batch & b = mod->impl->batch(id);
      need(which, "which"); need(out, "out");
      if (std::string(which) == "phase")
      {
         if (cap < (size_t) b.n_runs * 8) throw std::runtime_error("buffer too small!");
         std::vector<long long> tmp((size_t) b.n_runs * 8);
         b.get_phase_cycles(tmp.data());
         for (size_t i=0; i<tmp.size(); i++) out[i] = (double) tmp[i];
         return;
      }
      if (std::string(which) == "plan")
      {
         if (cap < 8) throw std::runtime_error("buffer too small!");
         b.get_plan(out);
         return;
      }
      if (cap < (size_t) b.n_runs * b.m * b.n) throw std::runtime_error("buffer too small!");
      b.get_state(which, out);
   });
}

int orc_batch_dims(orc_module * mod, int id, int * n_runs, int * n_points, int * n)
{
   return guarded(mod, [&] {
      orc::Batch & b = mod->impl->batch(id);
      if (n_runs) *n_runs = b.n_runs;
      if (n_points) *n_points = b.n_points;
      if (n) *n = b.n;
   });
}

int orc_batch_set_traj(orc_module * mod, int id, const double * traj, size_t count)
{
   return guarded(mod, [&] {
      orc::Batch & b = mod->impl->batch(id);
      need(traj, "traj");
      if (count != (size_t) b.n_runs * b.n_points * b.n) throw std::runtime_error("wrong trajectory size!");
      b.set_traj(traj);
   });
}

const char * orc_last_collision_details(const orc_module * mod)
{
   return mod ? mod->impl->last_collision_details.c_str() : "";
}

int orc_batch_destroy(orc_module * mod, int id)
{
   return guarded(mod, [&] { mod->impl->destroy_batch(id); });
}

int orc_kernel_time(orc_module * mod, double * total_ms, int * launches, int reset)
{
   return guarded(mod, [&] {
      mod->impl->time_collect();
      if (total_ms) *total_ms = mod->impl->kernel_ms_total;
      if (launches) *launches = mod->impl->kernel_launches;
      if (reset) { mod->impl->kernel_ms_total = 0.0; mod->impl->kernel_launches = 0; }
   });
}

int orc_host_bin_sdf(const int sizes[3], const double lengths[3], const double * occupancy, double * sdf_out)
{
   try
   {
      orc::Grid occ, sdf;
      for (int i=0; i<3; i++) { occ.sizes[i] = sizes[i]; occ.lengths[i] = lengths[i]; }
      occ.data.assign(occupancy, occupancy + occ.ncells());
      orc::grid_bin_sdf(occ, sdf);
      std::memcpy(sdf_out, sdf.data.data(), sdf.ncells() * sizeof(double));
      return 0;
   }
   catch (...) { return 1; }
}

int orc_host_flood_fill(const int sizes[3], double * cells, size_t start)
{
   try
   {
      orc::Grid g;
      for (int i=0; i<3; i++) { g.sizes[i] = sizes[i]; g.lengths[i] = 1.0; }
      g.data.assign(cells, cells + g.ncells());
      orc::grid_flood_1_to_0(g, start);
      std::memcpy(cells, g.data.data(), g.ncells() * sizeof(double));
      return 0;
   }
   catch (...) { return 1; }
}

int orc_host_voxelize_boxes(const int sizes[3], const double lengths[3], const double pose_world_gsdf[7], double cube_extent,
   int n_boxes, const double * box_world_poses, const double * half_extents, double * occupancy_out)
{
   try
   {
      orc::Grid g;
      for (int i=0; i<3; i++) { g.sizes[i] = sizes[i]; g.lengths[i] = lengths[i]; }
      std::vector<orc::Box> boxes(n_boxes);
      for (int k=0; k<n_boxes; k++)
      {
         boxes[k].world = orc::xform_from_pose(orc::Pose(box_world_poses + 7*k));
         for (int q=0; q<3; q++) boxes[k].half[q] = half_extents[3*k+q];
      }
      orc::voxelize_boxes(g, orc::Pose(pose_world_gsdf), cube_extent, boxes);
      std::memcpy(occupancy_out, g.data.data(), g.ncells() * sizeof(double));
      return 0;
   }
   catch (...) { return 1; }
}

int orc_host_voxelize_trimesh(const int sizes[3], const double lengths[3], const double pose_world_gsdf[7], double cube_extent,
   int n_tri, const double * world_vertices, double * occupancy_out)
{
   try
   {
      if (!sizes || !lengths || !pose_world_gsdf || !occupancy_out || (n_tri > 0 && !world_vertices) || n_tri > (1 << 26)) return 1;
      for (int i=0; i<3; i++) if (sizes[i] < 1 || !(lengths[i] > 0.0)) return 1;
      if (!(cube_extent > 0.0)) return 1;
      orc::Grid g;
      for (int i=0; i<3; i++) { g.sizes[i] = sizes[i]; g.lengths[i] = lengths[i]; }
      const std::vector<double> tris(world_vertices, world_vertices + 9*(size_t)(n_tri > 0 ? n_tri : 0));
      orc::voxelize_boxes(g, orc::Pose(pose_world_gsdf), cube_extent, std::vector<orc::Box>(), tris);
      std::memcpy(occupancy_out, g.data.data(), g.ncells() * sizeof(double));
      return 0;
   }
   catch (...) { return 1; }
}

int orc_host_shparse(const char * in, char * out, size_t out_cap)
{
   const std::vector<std::string> toks = orc::shparse(in ? in : "");
   size_t need = 0;
   for (const std::string & t : toks) need += t.size() + 1;
   if (need > out_cap) return -1;
   size_t off = 0;
   for (const std::string & t : toks) { std::memcpy(out + off, t.c_str(), t.size() + 1); off += t.size() + 1; }
   return (int) toks.size();
}

static int host_metric_impl(bool free_start, int m, int derivative, double dt, double * A_out, double * beta_s_out, double * beta_g_out,
   double kappa_out[3], const double * rhs, int ncols, double * solve_out)
{
   try
   {
      orc::Metric M;
      orc::build_metric(m, derivative, dt, M, free_start);
      if (A_out) std::memcpy(A_out, M.Adense.data(), (size_t) m*m*sizeof(double));
      if (beta_s_out) std::memcpy(beta_s_out, M.beta_s.data(), m*sizeof(double));
      if (beta_g_out) std::memcpy(beta_g_out, M.beta_g.data(), m*sizeof(double));
      if (kappa_out) { kappa_out[0] = M.kss; kappa_out[1] = M.ksg; kappa_out[2] = M.kgg; }
      if (rhs && solve_out)
      {
         const int n = ncols;
         if (derivative == 1)
         {
            // the device's cyclic reduction, executed serially with the same tables
            std::vector<double> cur(rhs, rhs + (size_t) m*n), nxt((size_t) m*n);
            int stride = 1;
            for (int l=0; l<M.pcr_levels; l++)
            {
               const double * ka = &M.pcr[(size_t)(2*l)*m]; const double * kc = ka + m;
               for (int i=0; i<m; i++) for (int c=0; c<n; c++)
               {
                  double d = cur[(size_t) i*n+c];
                  if (i-stride >= 0) d += ka[i] * cur[(size_t)(i-stride)*n+c];
                  if (i+stride < m)  d += kc[i] * cur[(size_t)(i+stride)*n+c];
                  nxt[(size_t) i*n+c] = d;
               }
               cur.swap(nxt);
               stride <<= 1;
            }
            const double * invb = &M.pcr[(size_t)(2*M.pcr_levels)*m];
            for (int i=0; i<m; i++) for (int c=0; c<n; c++) solve_out[(size_t) i*n+c] = cur[(size_t) i*n+c] * invb[i];
         }
         else if (M.ss_rank > 0)
            orc::semisep_apply(M, rhs, n, solve_out);      // the device's scans over the band inverse's generators, serially
         else
            for (int i=0; i<m; i++) for (int c=0; c<n; c++)
            {
               double s = 0.0;
               for (int k=0; k<m; k++) s += M.Ainv[(size_t) i*m+k] * rhs[(size_t) k*n+c];
               solve_out[(size_t) i*n+c] = s;
            }
      }
      return 0;
   }
   catch (...) { return 1; }
}

int orc_host_metric(int m, int derivative, double dt, double * A_out, double * beta_s_out, double * beta_g_out,
   double kappa_out[3], const double * rhs, int ncols, double * solve_out)
{
   return host_metric_impl(false, m, derivative, dt, A_out, beta_s_out, beta_g_out, kappa_out, rhs, ncols, solve_out);
}
int orc_host_metric_free_start(int m, int derivative, double dt, double * A_out, double * beta_s_out, double * beta_g_out,
   double kappa_out[3], const double * rhs, int ncols, double * solve_out)
{
   return host_metric_impl(true, m, derivative, dt, A_out, beta_s_out, beta_g_out, kappa_out, rhs, ncols, solve_out);
}

int orc_host_metric_semisep_rank(int m, int derivative, double dt, int free_start)
{
   try
   {
      if (m < 1 || derivative < 1) return -1;
      orc::Metric M;
      orc::build_metric(m, derivative, dt, M, free_start != 0);
      return M.ss_rank;
   }
   catch (...) { return -1; }
}

int orc_host_gsl_stream(unsigned long seed, double sigma, int n, double * out_gauss, double * out_uniform)
{
   orc::GslRng r(seed);
   for (int i=0; i<n; i++) out_gauss[i] = r.gaussian(sigma);
   if (out_uniform) out_uniform[0] = r.uniform();
   return 0;
}

} // extern "C"
