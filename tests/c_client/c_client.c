/* c_client.c -- a plain C99 client of the C ABI (include/orcdchomp_amd.h), the way a maintainer of the
 * reference would call it from C: a 3-link planar arm, one box obstacle, computedistancefield through
 * the command surface, a batch of runs through the kernel-level entry points, and the same run once
 * more through the reference's own create / iterate / gettraj / destroy command strings.
 * Prints one line per check; exit status 0 when everything agrees.  Built and run by
 * tests/test_gpu_c_client.py (gcc, no C++ runtime, no Python). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "orcdchomp_amd.h"

#define CHECK(call) do { if ((call) != 0) { fprintf(stderr, "FAILED %s: %s\n", #call, orc_last_error(m)); return 1; } } while (0)

int main(void)
{
   orc_module * m = orc_module_new(0);
   if (!m) { fprintf(stderr, "no module: %s\n", orc_last_error(NULL)); return 1; }

   /* robot: base + three revolute links about y, spheres along x */
   enum { NL = 4, NS = 5 };
   const int parent[NL] = { -1, 0, 1, 2 };
   const double ppj[NL][7] = { {0,0,0, 0,0,0,1}, {0,0,0.9, 0,0,0,1}, {0.35,0,0, 0,0,0,1}, {0.3,0,0, 0,0,0,1} };
   const int jtype[NL] = { 0, 1, 1, 1 };
   const double axis[NL][3] = { {0,0,1}, {0,1,0}, {0,1,0}, {0,1,0} };
   const int dof[NL] = { -1, 0, 1, 2 };
   const double lo[3] = { -1.5, -2.2, -2.2 }, hi[3] = { 1.5, 2.2, 2.2 };
   const int slink[NS] = { 1, 1, 2, 2, 3 };
   const double spos[NS][3] = { {0.1,0,0}, {0.25,0,0}, {0.1,0,0}, {0.22,0,0}, {0.12,0,0} };
   const double srad[NS] = { 0.06, 0.06, 0.05, 0.05, 0.05 };
   orc_robot_desc d;
   d.n_links = NL; d.parent = parent; d.pose_parent_joint = &ppj[0][0]; d.joint_type = jtype; d.axis = &axis[0][0];
   d.dof_index = dof; d.n_dof = 3; d.limit_lower = lo; d.limit_upper = hi;
   d.n_spheres = NS; d.sphere_link = slink; d.sphere_pos = &spos[0][0]; d.sphere_radius = srad;
   CHECK(orc_env_add_robot(m, "arm3", &d));
   const double base[7] = { 0,0,0, 0,0,0,1 };
   const double q0[3] = { -1.0, 0.4, 0.3 };
   const int adofs[3] = { 0, 1, 2 };
   CHECK(orc_robot_set_transform(m, "arm3", base));
   CHECK(orc_robot_set_dof_values(m, "arm3", q0, 3));
   CHECK(orc_robot_set_active_dofs(m, "arm3", adofs, 3));

   /* obstacle: a box in front of the arm, and its signed distance field */
   const double bpose[7] = { 0.45, 0.0, 0.75, 0,0,0,1 }, bhalf[3] = { 0.08, 0.3, 0.08 };
   CHECK(orc_env_add_kinbody_boxes(m, "bar", 1, bpose, bhalf));
   char reply[256];
   CHECK(orc_send_command(m, "computedistancefield kinbody bar cube_extent 0.02 aabb_padding 0.3", reply, sizeof reply));
   int sizes[3]; double lengths[3], gpose[7];
   CHECK(orc_scene_get_sdf(m, "bar", sizes, lengths, gpose, NULL, 0));
   printf("field of 'bar': %d x %d x %d cells, %.2f x %.2f x %.2f m\n", sizes[0], sizes[1], sizes[2], lengths[0], lengths[1], lengths[2]);

   /* a batch of 8 runs through the kernel-level entry points */
   enum { NR = 8, NP = 40 };
   orc_batch_params p; orc_batch_params_default(&p);
   p.n_points = NP; p.lambda = 50.0; p.obs_factor = 300.0;
   double goals[NR][3];
   for (int k=0; k<NR; k++) { goals[k][0] = 0.9 - 0.05*k; goals[k][1] = -0.5 + 0.1*k; goals[k][2] = 0.2; }
   int bid = 0;
   CHECK(orc_batch_create(m, "arm3", &p, NR, NULL, &goals[0][0], NULL, NULL, &bid));
   double costs[NR][3]; int status[NR];
   CHECK(orc_batch_iterate(m, bid, 60, &costs[0][0], status));
   static double traj[NR][NP][3];
   CHECK(orc_batch_gettraj(m, bid, &traj[0][0][0], (size_t) NR*NP*3));
   int bad = 0;
   for (int k=0; k<NR; k++)
   {
      if (status[k] != 0) bad++;
      for (int j=0; j<3; j++)
      {
         if (traj[k][0][j] != q0[j]) bad++;                                  /* the start never moves */
         if (fabs(traj[k][NP-1][j] - goals[k][j]) > 1e-12) bad++;             /* nor does the goal */
      }
      if (!(fabs(costs[k][0] - (costs[k][1] + costs[k][2])) <= 1e-9 * fabs(costs[k][0]))) bad++;
   }
   printf("batch of %d runs: cost of run 0 = %.6f (obstacle %.6f + smoothness %.6f), inconsistencies: %d\n", NR, costs[0][0], costs[0][1], costs[0][2], bad);

   /* run 3 once more through the reference's command strings */
   char cmd[512], handle[64];
   snprintf(cmd, sizeof cmd, "create robot arm3 adofgoal '%.17g %.17g %.17g' lambda 50.0000 n_points %d obs_factor 300.000000",
            goals[3][0], goals[3][1], goals[3][2], NP);
   CHECK(orc_send_command(m, cmd, handle, sizeof handle));
   snprintf(cmd, sizeof cmd, "iterate run %s n_iter 60", handle);
   CHECK(orc_send_command(m, cmd, reply, sizeof reply));
   const double cost_cmd = atof(reply);
   printf("the same run through create/iterate: cost %.6f vs %.6f in the batch\n", cost_cmd, costs[3][0]);
   if (!(fabs(cost_cmd - costs[3][0]) <= 1e-5 * fabs(costs[3][0]))) bad++;    /* the reply is printed with 6 digits */
   snprintf(cmd, sizeof cmd, "gettraj run %s no_collision_check", handle);
   CHECK(orc_send_command(m, cmd, reply, sizeof reply));
   size_t full = orc_last_reply_size(m);
   char * doc = (char *) malloc(full + 1);
   CHECK(orc_last_reply(m, doc, full + 1));
   if (!strstr(doc, "<trajectory>") && !strstr(doc, "<trajectory")) bad++;
   printf("gettraj: %zu bytes of trajectory document\n", full);
   free(doc);
   snprintf(cmd, sizeof cmd, "destroy run %s", handle);
   CHECK(orc_send_command(m, cmd, reply, sizeof reply));

   /* errors come back with the reference's messages */
   if (orc_send_command(m, "create robot arm3", reply, sizeof reply) == 0) bad++;
   else printf("error path: \"%s\"\n", orc_last_error(m));
   if (strcmp(orc_last_error(m), "Did not pass either adofgoal or starttraj!") != 0) bad++;

   /* a TSR hard constraint through the command layer: keep the tool's height (link names and a
    * manipulator are what `con_tsr 'all manipee NAME'` addresses; identity frames, z row fixed) */
   {
      const char * names[NL] = { "base", "upper", "fore", "hand" };
      const double tool[7] = { 0.15, 0, 0, 0, 0, 0, 1 };
      CHECK(orc_robot_set_link_names(m, "arm3", names, NL));
      CHECK(orc_robot_add_manipulator(m, "arm3", "gripper", 3, tool));
      CHECK(orc_robot_set_active_manipulator(m, "arm3", "gripper"));
      CHECK(orc_set_workgroup_threads(m, 0));
      snprintf(cmd, sizeof cmd, "create robot arm3 adofgoal '%.17g %.17g %.17g' lambda 50.0000 n_points %d obs_factor 300.000000 "
               "con_tsr 'all manipee gripper' '0 NULL 1 0 0 0 1 0 0 0 1 0 0 0  1 0 0 0 1 0 0 0 1 0 0 0  -9 9 -9 9 0 0 -9 9 -9 9 -9 9'",
               -1.0, 0.4, 0.3, NP);      /* goal = start: the constraint pins the tool at z = 0 of the world, which it is not at */
      CHECK(orc_send_command(m, cmd, handle, sizeof handle));
      snprintf(cmd, sizeof cmd, "iterate run %s n_iter 5", handle);
      CHECK(orc_send_command(m, cmd, reply, sizeof reply));
      printf("a TSR-constrained run through create/iterate: cost %s\n", reply);
      snprintf(cmd, sizeof cmd, "destroy run %s", handle);
      CHECK(orc_send_command(m, cmd, reply, sizeof reply));
      if (orc_send_command(m, "create robot arm3 adofgoal '0 0 0' con_tsr 'all link nope' '0 NULL 1 0 0 0 1 0 0 0 1 0 0 0  1 0 0 0 1 0 0 0 1 0 0 0  0 0 0 0 0 0 0 0 0 0 0 0'",
                           reply, sizeof reply) == 0 || strcmp(orc_last_error(m), "con_tsr link not found!") != 0) bad++;
   }

   /* the arm picks something up (RobotBase::Grab): a kinbody with <orcdchomp> spheres held by the hand link; create then
    * collects its spheres with the robot's (src/orcdchomp_mod.cpp:2168-2300), and released the robot plans as before */
   {
      const double part_pose[7] = { 0.5, 0.0, 1.4, 0,0,0,1 }, origin[7] = { 0,0,0, 0,0,0,1 }, part_half[3] = { 0.03, 0.03, 0.03 };
      const double part_sph[2][3] = { {0,0,0}, {0.07,0,0} }, part_rad[2] = { 0.04, 0.03 };
      double costs_held[NR][3], costs_free[NR][3], where[7];
      int bid_held = 0, bid_free = 0;
      CHECK(orc_env_add_kinbody_boxes(m, "part", 1, origin, part_half));
      CHECK(orc_kinbody_set_transform(m, "part", part_pose));                   /* next to the hand at q0 */
      CHECK(orc_kinbody_set_spheres(m, "part", 2, &part_sph[0][0], part_rad));
      CHECK(orc_robot_grab(m, "arm3", "part", 3));
      CHECK(orc_batch_create(m, "arm3", &p, NR, NULL, &goals[0][0], NULL, NULL, &bid_held));
      CHECK(orc_batch_iterate(m, bid_held, 60, &costs_held[0][0], status));
      CHECK(orc_body_get_transform(m, "part", where));
      CHECK(orc_robot_release(m, "arm3", "part"));
      CHECK(orc_batch_create(m, "arm3", &p, NR, NULL, &goals[0][0], NULL, NULL, &bid_free));
      CHECK(orc_batch_iterate(m, bid_free, 60, &costs_free[0][0], status));
      int differ = 0;
      for (int k=0; k<NR; k++)
      {
         if (costs_free[k][0] != costs[k][0]) bad++;                          /* released: the bits of the robot that never held anything */
         if (costs_held[k][0] != costs[k][0]) differ++;
      }
      if (!differ || fabs(where[0] - 0.5) > 1e-12) bad++;
      printf("holding 'part' (2 spheres on the hand): cost of run 0 %.6f against %.6f without; the part sits at x = %.3f\n", costs_held[0][0], costs[0][0], where[0]);
      CHECK(orc_batch_destroy(m, bid_held));
      CHECK(orc_batch_destroy(m, bid_free));
      if (orc_robot_release(m, "arm3", "part") == 0 || !strstr(orc_last_error(m), "not grabbing")) bad++;
   }

   int collides[NR];
   CHECK(orc_batch_collision_verdict(m, bid, collides, NULL, NULL, NULL, NULL));
   int nc = 0; for (int k=0; k<NR; k++) nc += collides[k];
   printf("collision verdict: %d of %d trajectories touch the bar\n", nc, NR);
   CHECK(orc_batch_destroy(m, bid));
   orc_module_free(m);
   printf(bad ? "C CLIENT FAILED (%d)\n" : "C CLIENT OK\n", bad);
   return bad ? 1 : 0;
}
