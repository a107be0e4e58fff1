/* ora_run.c -- TEST INFRASTRUCTURE (see oracle.h).
 * CPU restatement of the orcdchomp module's run lifecycle and obstacle-cost
 * callbacks (src/orcdchomp_mod.cpp:850-1327, 1800-2852), with the build's own
 * kinematic model standing in for OpenRAVE FK / CalculateJacobian (third party,
 * absent: PARITY UNPINNED for FK and Jacobians, SURVEY 2 last table).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle.h"

static void self_pairs(const ora_robot * rob, int ns, const int * link, const double * pos, const double * radius, unsigned char * excl);

/* ------------------------------------------------------------- robot FK */

static void mat3_mul(const double * A, const double * B, double * C)
{
   int i, j, k;
   for (i=0; i<3; i++) for (j=0; j<3; j++)
   {
      double s = 0.0;
      for (k=0; k<3; k++) s += A[i*3+k] * B[k*3+j];
      C[i*3+j] = s;
   }
}

static void mat3_vec(const double * A, const double * v, double * out)
{
   int i;
   for (i=0; i<3; i++) out[i] = A[i*3+0]*v[0] + A[i*3+1]*v[1] + A[i*3+2]*v[2];
}

/* Rodrigues rotation about a unit axis */
static void axis_angle_R(const double * a, double q, double * R)
{
   double c = cos(q), s = sin(q), v = 1.0 - c;
   R[0] = c + a[0]*a[0]*v;      R[1] = a[0]*a[1]*v - a[2]*s; R[2] = a[0]*a[2]*v + a[1]*s;
   R[3] = a[1]*a[0]*v + a[2]*s; R[4] = c + a[1]*a[1]*v;      R[5] = a[1]*a[2]*v - a[0]*s;
   R[6] = a[2]*a[0]*v - a[1]*s; R[7] = a[2]*a[1]*v + a[0]*s; R[8] = c + a[2]*a[2]*v;
}

void ora_robot_fk(const ora_robot * rob, const double base_pose[7], const double * dofvals,
   double * R, double * t, double * axis_w, double * anchor_w)
{
   int li;
   double Rb[3][3];
   ora_kin_quat_to_R(base_pose+3, Rb);
   for (li=0; li<rob->n_links; li++)
   {
      const double * pj = rob->pose_parent_joint + 7*li;
      const double * Rp; const double * tp;
      double Rfix[3][3], Rj[9], tj[3], Rm[9];
      int type = rob->joint_type[li];
      int dof = rob->dof_index[li];
      double q = (dof >= 0 && type != 0) ? dofvals[dof] : 0.0;
      if (rob->parent[li] < 0) { Rp = &Rb[0][0]; tp = base_pose; }
      else { Rp = R + 9*rob->parent[li]; tp = t + 3*rob->parent[li]; }
      /* joint frame in the world */
      ora_kin_quat_to_R(pj+3, Rfix);
      mat3_mul(Rp, &Rfix[0][0], Rj);
      mat3_vec(Rp, pj, tj);
      tj[0] += tp[0]; tj[1] += tp[1]; tj[2] += tp[2];
      if (axis_w) mat3_vec(Rj, rob->axis + 3*li, axis_w + 3*li);
      if (anchor_w) { anchor_w[3*li+0] = tj[0]; anchor_w[3*li+1] = tj[1]; anchor_w[3*li+2] = tj[2]; }
      if (type == 1)
      {
         axis_angle_R(rob->axis + 3*li, q, Rm);
         mat3_mul(Rj, Rm, R + 9*li);
         t[3*li+0] = tj[0]; t[3*li+1] = tj[1]; t[3*li+2] = tj[2];
      }
      else
      {
         memcpy(R + 9*li, Rj, 9*sizeof(double));
         if (type == 2)
         {
            double aw[3];
            mat3_vec(Rj, rob->axis + 3*li, aw);
            tj[0] += q*aw[0]; tj[1] += q*aw[1]; tj[2] += q*aw[2];
         }
         else if (axis_w) { axis_w[3*li+0] = 0.0; axis_w[3*li+1] = 0.0; axis_w[3*li+2] = 0.0; }
         t[3*li+0] = tj[0]; t[3*li+1] = tj[1]; t[3*li+2] = tj[2];
      }
   }
}

int ora_robot_does_affect(const ora_robot * rob, int dof, int link)
{
   int li;
   for (li=link; li>=0; li=rob->parent[li])
      if (rob->joint_type[li] != 0 && rob->dof_index[li] == dof) return 1;
   return 0;
}

/* ------------------------------------------------------------------ run */

typedef struct run_sphere   /* src/orcdchomp_mod.cpp:857-864 */
{
   double radius;
   int robot_linkindex;
   double pos_wrt_link[3];
   int xml_index;
} run_sphere;

struct ora_run              /* src/orcdchomp_mod.cpp:887-966 */
{
   double * traj;
   int n_points;
   const ora_robot * robot;
   double base_pose[7];
   double * dofvals;        /* all robot dofs; inactive ones stay at their create-time values */
   int n_adof;
   int floating_base;
   int * adofindices;
   double epsilon, epsilon_self, obs_factor, obs_factor_self;
   run_sphere * spheres;    /* array, first n_spheres_active are active */
   int n_spheres, n_spheres_active;
   /* the spheres create collected, robot first, then the grabbed bodies in GetGrabbed() order, each in XML order
    * (xml_index counts through this list); a grabbed body's spheres sit on the grabbing link */
   int * eff_link; double * eff_pos; double * eff_radius;
   unsigned char * self_excl;   /* [n_eff][n_eff] pairs of them the re-check's self-collision leg never tests, taken at create (run_self_pairs) */
   int n_eff;
   double * sphere_poss_inactive;
   double * sphere_poss_all;
   double * sphere_poss;
   double * sphere_vels;
   double * sphere_accs;
   double * sphere_jacs;
   double * J2;
   int n_rsdfs;
   ora_rsdf * rsdfs;
   int use_hmc;
   int hmc_resample_iter;
   double hmc_resample_lambda;
   ora_rng rng;
   ora_chomp * c;
   int iter;
   /* FK scratch */
   double * fkR, * fkt, * fkaxis, * fkanchor;
   /* tsr constraints applied (struct run_contsr, src/orcdchomp_mod.cpp:873-885) */
   int n_contsrs;
   struct run_contsr ** contsrs;
   struct run_contsr * start_contsr;   /* start_tsr */
};

typedef struct run_contsr
{
   ora_run * r;
   int ee_link; double tool[7];      /* manip->GetEndEffectorTransform() = link transform o tool; identity for a link */
   double T0w[7], Twe[7], Bw[6][2];  /* struct tsr, src/orcdchomp_mod.h:80-87 */
   int tsr_enabled[6];               /* xyzrpy */
   int k;
} run_contsr;

void ora_run_params_default(ora_run_params * p)   /* src/orcdchomp_mod.cpp:1824-1826,1838-1885 */
{
   p->n_points = 101;
   p->floating_base = 0;
   p->lambda = 10.0;
   p->D = 1;
   p->use_momentum = 0;
   p->use_hmc = 0;
   p->hmc_resample_lambda = 0.02;
   p->seed = 0;
   p->epsilon = 0.1;
   p->epsilon_self = 0.04;
   p->obs_factor = 200.0;
   p->obs_factor_self = 10.0;
   p->start_tsr = 0;
}

static double nrm2_3(const double * v) { return sqrt(v[0]*v[0] + v[1]*v[1] + v[2]*v[2]); }
static double dot3(const double * a, const double * b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }

/* src/orcdchomp_mod.cpp:968-1132 */
static int sphere_cost_pre(void * vptr, ora_chomp * c, int m, double ** T_points)
{
   ora_run * r = (ora_run *) vptr;
   const ora_robot * rob = r->robot;
   int Sa = r->n_spheres_active, n = c->n;
   int ti, sai, i, j, k;
   (void) m; (void) T_points;

   for (ti=0; ti<r->n_points; ti++)
   {
      const double * row = &r->traj[ti*n];
      double pose[7];
      double Jsp[6][7];
      int ti_mov;
      /* put the robot in the config (mod.cpp:991-1028) */
      if (r->floating_base)
      {
         ora_spatial_pose_jac(row, Jsp);
         for (i=0; i<7; i++) pose[i] = row[i];
         for (j=0; j<r->n_adof; j++) r->dofvals[r->adofindices[j]] = row[7+j];
      }
      else
      {
         for (i=0; i<7; i++) pose[i] = r->base_pose[i];
         for (j=0; j<r->n_adof; j++) r->dofvals[r->adofindices[j]] = row[j];
      }
      ora_robot_fk(rob, pose, r->dofvals, r->fkR, r->fkt, r->fkaxis, r->fkanchor);

      ti_mov = (c->m == r->n_points - 2) ? ti - 1 : ti;   /* mod.cpp:1040-1043 */
      for (sai=0; sai<Sa; sai++)
      {
         const run_sphere * s = &r->spheres[sai];
         int li = s->robot_linkindex;
         double v[3];
         double * jac;
         mat3_vec(r->fkR + 9*li, s->pos_wrt_link, v);
         v[0] += r->fkt[3*li+0]; v[1] += r->fkt[3*li+1]; v[2] += r->fkt[3*li+2];
         r->sphere_poss_all[ti*(Sa*3) + sai*3 + 0] = v[0];
         r->sphere_poss_all[ti*(Sa*3) + sai*3 + 1] = v[1];
         r->sphere_poss_all[ti*(Sa*3) + sai*3 + 2] = v[2];
         if (ti_mov < 0 || c->m <= ti_mov) continue;
         jac = r->sphere_jacs + ti_mov*Sa*3*n + sai*3*n;
         memset(jac, 0, 3*n*sizeof(double));
         /* linear Jacobian of the sphere centre wrt the active dofs:
          * CalculateJacobian semantics */
         for (j=0; j<r->n_adof; j++)
         {
            int dof = r->adofindices[j];
            int col = (r->floating_base ? 7 : 0) + j;
            int lk;
            for (lk=li; lk>=0; lk=rob->parent[lk])
            {
               const double * a = r->fkaxis + 3*lk;
               if (rob->joint_type[lk] == 0 || rob->dof_index[lk] != dof) continue;
               if (rob->joint_type[lk] == 1)
               {
                  double d[3];
                  d[0] = v[0] - r->fkanchor[3*lk+0];
                  d[1] = v[1] - r->fkanchor[3*lk+1];
                  d[2] = v[2] - r->fkanchor[3*lk+2];
                  jac[0*n+col] += a[1]*d[2] - a[2]*d[1];
                  jac[1*n+col] += a[2]*d[0] - a[0]*d[2];
                  jac[2*n+col] += a[0]*d[1] - a[1]*d[0];
               }
               else
               {
                  jac[0*n+col] += a[0]; jac[1*n+col] += a[1]; jac[2*n+col] += a[2];
               }
            }
         }
         if (r->floating_base)
         {
            /* mod.cpp:1050-1080: left 3x7 block = Xm[3:6,:] * Jsp, then *= 0.01 */
            double spose[7], Xm[6][6];
            ora_kin_pose_identity(spose);
            spose[0] = -v[0]; spose[1] = -v[1]; spose[2] = -v[2];
            ora_spatial_xm_from_pose(Xm, spose);
            for (i=0; i<3; i++) for (j=0; j<7; j++)
            {
               double sum = 0.0;
               for (k=0; k<6; k++) sum += Xm[3+i][k] * Jsp[k][j];
               jac[i*n+j] = sum;
            }
            for (i=0; i<3; i++) for (j=0; j<7; j++) jac[i*n+j] *= 0.01;
         }
      }
   }

   /* central-difference sphere velocities and accelerations (mod.cpp:1099-1127); with start_tsr the
    * internal points follow the start point, whose velocity is one-sided and whose acceleration is
    * the first internal one */
   {
      int rows = r->n_points - 2, cols = Sa*3;
      const double * P = r->sphere_poss_all;
      double * vels = r->sphere_vels, * accs = r->sphere_accs;
      if (c->m != r->n_points - 2) { vels += cols; accs += cols; }
      for (i=0; i<rows; i++) for (j=0; j<cols; j++)
      {
         double vel = P[(i+2)*cols+j];
         double acc = P[(i+1)*cols+j];
         vel -= P[i*cols+j];
         vel *= 1.0/(2.0*c->dt);
         acc *= -2.0;
         acc += P[i*cols+j];
         acc += P[(i+2)*cols+j];
         acc *= 1.0/(c->dt * c->dt);
         vels[i*cols+j] = vel;
         accs[i*cols+j] = acc;
      }
      if (c->m != r->n_points - 2)
         for (j=0; j<cols; j++)
         {
            double vel = P[cols+j];
            vel -= P[j];
            vel *= 1.0/(c->dt);
            r->sphere_vels[j] = vel;
            r->sphere_accs[j] = r->sphere_accs[cols+j];
         }
   }
   return 0;
}

/* src/orcdchomp_mod.cpp:1134-1327 */
static int sphere_cost(void * vptr, ora_chomp * c, int ti, double * c_point, double * c_vel, double * costp, double * c_grad)
{
   ora_run * r = (ora_run *) vptr;
   int Sa = r->n_spheres_active, n = c->n;
   int sai, sai2, i, j;
   double cost = 0.0;
   (void) c_point; (void) c_vel;

   if (c_grad) memset(c_grad, 0, n*sizeof(double));

   for (sai=0; sai<Sa; sai++)
   {
      const run_sphere * sact = &r->spheres[sai];
      const double * x_vel = r->sphere_vels + ti*(Sa*3) + sai*3;
      const double * pos = r->sphere_poss + ti*(Sa*3) + sai*3;
      const double * jac = r->sphere_jacs + ti*Sa*3*n + sai*3*n;
      double x_vel_norm = nrm2_3(x_vel);
      double cost_sphere = 0.0;
      double g_point[3], g_grad[3], x_grad[3], x_curv[3], dist, proj;
      double best = HUGE_VAL;
      int best_i = -1;

      /* field with the smallest value (mod.cpp:1171-1188) */
      for (i=0; i<r->n_rsdfs; i++)
      {
         ora_kin_pose_compos(r->rsdfs[i].pose_gsdf_world, pos, g_point);
         if (ora_grid_double_interp(r->rsdfs[i].grid, g_point, &dist)) continue;
         if (dist < best) { best = dist; best_i = i; }
      }
      if (best_i != -1)
      {
         ora_kin_pose_compos(r->rsdfs[best_i].pose_gsdf_world, pos, g_point);
         ora_grid_double_interp(r->rsdfs[best_i].grid, g_point, &dist);
         dist -= sact->radius;
         if (dist < 0.0)
            cost_sphere += x_vel_norm * r->obs_factor * (0.5 * r->epsilon - dist);
         else if (dist < r->epsilon)
            cost_sphere += x_vel_norm * r->obs_factor * (0.5/r->epsilon) * (dist - r->epsilon) * (dist - r->epsilon);

         if (c_grad)
         {
            ora_grid_double_grad(r->rsdfs[best_i].grid, g_point, g_grad);
            ora_kin_pose_compose_vec(r->rsdfs[best_i].pose_world_gsdf, g_grad, g_grad);
            for (j=0; j<3; j++) x_grad[j] = g_grad[j];
            if (dist < 0.0)
               for (j=0; j<3; j++) x_grad[j] *= -1.0;
            else if (dist < r->epsilon)
               for (j=0; j<3; j++) x_grad[j] *= dist/r->epsilon - 1.0;
            else
               for (j=0; j<3; j++) x_grad[j] = 0.0;
            for (j=0; j<3; j++) x_grad[j] *= x_vel_norm * r->obs_factor;
            if (x_vel_norm > 0.000001)
            {
               proj = dot3(x_grad, x_vel) / (x_vel_norm * x_vel_norm);
               for (j=0; j<3; j++) x_grad[j] += -proj * x_vel[j];
            }
            for (j=0; j<3; j++) x_curv[j] = r->sphere_accs[ti*(Sa*3) + sai*3 + j];
            if (x_vel_norm > 0.000001)
            {
               proj = dot3(x_curv, x_vel) / (x_vel_norm * x_vel_norm);
               for (j=0; j<3; j++) x_curv[j] += -proj * x_vel[j];
            }
            for (j=0; j<3; j++) x_curv[j] *= 1.0 / (x_vel_norm * x_vel_norm);
            for (j=0; j<3; j++) x_grad[j] += -cost_sphere * x_curv[j];
            /* dgemv(Trans, alpha=|v|): BLAS returns early for alpha==0, which is the
             * only thing that keeps the NaN in x_grad out of c_grad */
            if (x_vel_norm != 0.0)
               for (j=0; j<n; j++)
                  c_grad[j] += x_vel_norm * (jac[0*n+j]*x_grad[0] + jac[1*n+j]*x_grad[1] + jac[2*n+j]*x_grad[2]);
         }
      }

      /* self collision against every other sphere (mod.cpp:1251-1317) */
      for (sai2=0; sai2<r->n_spheres; sai2++)
      {
         const run_sphere * sact2 = &r->spheres[sai2];
         double v_from_other[3];
         if (sact->robot_linkindex == sact2->robot_linkindex) continue;
         for (j=0; j<3; j++) v_from_other[j] = pos[j];
         if (sai2 < Sa)
            for (j=0; j<3; j++) v_from_other[j] -= r->sphere_poss[ti*(Sa*3) + sai2*3 + j];
         else
            for (j=0; j<3; j++) v_from_other[j] -= r->sphere_poss_inactive[(sai2-Sa)*3 + j];
         dist = nrm2_3(v_from_other);
         if (dist > sact->radius + sact2->radius + r->epsilon_self) continue;
         if (c_grad)
            for (j=0; j<3; j++) g_grad[j] = v_from_other[j] / dist;
         dist -= sact->radius + sact2->radius;
         if (costp)
         {
            if (dist < 0.0)
               cost_sphere += x_vel_norm * r->obs_factor_self * (0.5 * r->epsilon_self - dist);
            else
               cost_sphere += x_vel_norm * r->obs_factor_self * (0.5/r->epsilon_self) * (dist - r->epsilon_self) * (dist - r->epsilon_self);
         }
         if (c_grad)
         {
            for (j=0; j<3; j++) x_grad[j] = g_grad[j];
            if (dist < 0.0)
               for (j=0; j<3; j++) x_grad[j] *= -1.0;
            else if (dist < r->epsilon_self)
               for (j=0; j<3; j++) x_grad[j] *= dist/r->epsilon_self - 1.0;
            for (j=0; j<3; j++) x_grad[j] *= x_vel_norm * r->obs_factor_self;
            if (x_vel_norm > 0.000001)
            {
               proj = dot3(x_grad, x_vel) / (x_vel_norm * x_vel_norm);
               for (j=0; j<3; j++) x_grad[j] += -proj * x_vel[j];
            }
            memcpy(r->J2, jac, 3*n*sizeof(double));
            if (sai2 < Sa)
            {
               const double * jac2 = r->sphere_jacs + ti*Sa*3*n + sai2*3*n;
               for (j=0; j<3*n; j++) r->J2[j] -= jac2[j];
            }
            for (j=0; j<n; j++)
               c_grad[j] += 1.0 * (r->J2[0*n+j]*x_grad[0] + r->J2[1*n+j]*x_grad[1] + r->J2[2*n+j]*x_grad[2]);
         }
      }
      cost += cost_sphere;
   }
   if (costp) *costp = cost;
   return 0;
}

static void run_free(ora_run * r)
{
   if (!r) return;
   free(r->traj); free(r->dofvals); free(r->adofindices); free(r->spheres);
   free(r->eff_link); free(r->eff_pos); free(r->eff_radius); free(r->self_excl);
   free(r->sphere_poss_inactive); free(r->sphere_poss_all); free(r->sphere_vels);
   free(r->sphere_accs); free(r->sphere_jacs); free(r->J2); free(r->rsdfs);
   free(r->fkR); free(r->fkt); free(r->fkaxis); free(r->fkanchor);
   { int j; for (j=0; j<r->n_contsrs; j++) free(r->contsrs[j]); free(r->contsrs); free(r->start_contsr); }
   ora_chomp_free(r->c);
   free(r);
}

/* ------------------------------------------------------------------ TSR constraint
 * con_tsr, src/orcdchomp_mod.cpp:1330-1497 (con_everyn_tsr, 1500-1657, is the same body with the
 * active manipulator's end effector).  The end-effector transform and the two OpenRAVE Jacobians
 * (CalculateAngularVelocityJacobian; CalculateJacobian at the world origin, i.e. the spatial
 * velocity convention) come from the build's own kinematic model. */
static void small_gemm(int M, int N, int K, const double * A, int lda, const double * B, int ldb, double * C, int ldc)
{
   int i, j, k;
   for (i=0; i<M; i++) for (j=0; j<N; j++)
   {
      double sum = 0.0;
      for (k=0; k<K; k++) sum += A[i*lda+k] * B[k*ldb+j];
      C[i*ldc+j] = sum;
   }
}

static int con_tsr(void * vptr, ora_chomp * c, int ti, double * point, double * con_val, double * con_jacobian)
{
   run_contsr * contsr = (run_contsr *) vptr;
   ora_run * r = contsr->r;
   const ora_robot * rob = r->robot;
   int n = c->n, tsri, ki, i, j;
   double base[7], pose_link[7], pose_ee[7], pose_obj[7], pose_ee_obj[7], pose_table_world[7], pose_table_obj[7];
   double xyzypr_table_obj[7];
   double Rl[3][3];
   (void) ti;

   /* put the arm in this configuration (mod.cpp:1357-1376) */
   if (r->floating_base)
   {
      for (i=0; i<7; i++) base[i] = point[i];
      for (j=0; j<r->n_adof; j++) r->dofvals[r->adofindices[j]] = point[7+j];
   }
   else
   {
      for (i=0; i<7; i++) base[i] = r->base_pose[i];
      for (j=0; j<r->n_adof; j++) r->dofvals[r->adofindices[j]] = point[j];
   }
   ora_robot_fk(rob, base, r->dofvals, r->fkR, r->fkt, r->fkaxis, r->fkanchor);

   /* the end-effector transform (mod.cpp:1382-1394) */
   for (i=0; i<3; i++) for (j=0; j<3; j++) Rl[i][j] = r->fkR[9*contsr->ee_link + 3*i + j];
   ora_kin_pose_from_dR(pose_link, r->fkt + 3*contsr->ee_link, Rl);
   ora_kin_pose_compose(pose_link, contsr->tool, pose_ee);

   /* object pose, world wrt the table, object wrt the table, as xyzypr (mod.cpp:1396-1407) */
   ora_kin_pose_invert(contsr->Twe, pose_ee_obj);
   ora_kin_pose_compose(pose_ee, pose_ee_obj, pose_obj);
   ora_kin_pose_invert(contsr->T0w, pose_table_world);
   ora_kin_pose_compose(pose_table_world, pose_obj, pose_table_obj);
   ora_kin_pose_to_xyzypr(pose_table_obj, xyzypr_table_obj);

   /* the constraint value vector (mod.cpp:1409-1415) */
   ki = 0;
   for (tsri=0; tsri<6; tsri++) if (contsr->tsr_enabled[tsri])
   {
      con_val[ki] = xyzypr_table_obj[tsri<3?tsri:8-tsri];
      ki++;
   }

   if (con_jacobian)
   {
      double * spajac_world = (double *) calloc((size_t) 6*n, sizeof(double));
      double * full_result = (double *) calloc((size_t) 6*n, sizeof(double));
      double Jsp[6][7], xm_table_world[6][6], jac_inverse[7][6], pose_to_xyzypr_jac[6][7];
      double temp6x6a[6][6], temp6x6b[6][6];
      int col0 = r->floating_base ? 7 : 0;
      if (r->floating_base)
      {
         /* floating base: first seven columns of the spatial velocity jacobian (mod.cpp:1434-1437) */
         ora_spatial_pose_jac(point, Jsp);
         for (i=0; i<6; i++) for (j=0; j<7; j++) spajac_world[i*n+j] = Jsp[i][j];
      }
      /* active columns: rows 0..2 the angular velocity Jacobian of the link, rows 3..5 the Jacobian
       * of the link's point that sits at the world origin (mod.cpp:1439-1464) */
      for (j=0; j<r->n_adof; j++)
      {
         int dof = r->adofindices[j], lj, found = -1;
         if (!ora_robot_does_affect(rob, dof, contsr->ee_link)) continue;
         for (lj=contsr->ee_link; lj>=0; lj=rob->parent[lj])
            if (rob->joint_type[lj] != 0 && rob->dof_index[lj] == dof) { found = lj; break; }
         if (found < 0) continue;
         {
            const double * ax = r->fkaxis + 3*found;
            const double * an = r->fkanchor + 3*found;
            if (rob->joint_type[found] == 1)
            {
               for (i=0; i<3; i++) spajac_world[i*n + col0+j] = ax[i];
               /* axis x (0 - anchor) */
               spajac_world[3*n + col0+j] = ax[1]*(-an[2]) - ax[2]*(-an[1]);
               spajac_world[4*n + col0+j] = ax[2]*(-an[0]) - ax[0]*(-an[2]);
               spajac_world[5*n + col0+j] = ax[0]*(-an[1]) - ax[1]*(-an[0]);
            }
            else
               for (i=0; i<3; i++) spajac_world[(3+i)*n + col0+j] = ax[i];
         }
      }
      /* velocities in the table frame, the pose derivative, the xyzypr Jacobian (mod.cpp:1466-1474) */
      ora_spatial_xm_from_pose(xm_table_world, pose_table_world);
      ora_spatial_pose_jac_inverse(pose_table_obj, jac_inverse);
      ora_kin_pose_to_xyzypr_J(pose_table_obj, pose_to_xyzypr_jac);
      /* the three products (mod.cpp:1476-1482) */
      small_gemm(6, 6, 7, &pose_to_xyzypr_jac[0][0], 7, &jac_inverse[0][0], 6, &temp6x6a[0][0], 6);
      small_gemm(6, 6, 6, &temp6x6a[0][0], 6, &xm_table_world[0][0], 6, &temp6x6b[0][0], 6);
      small_gemm(6, n, 6, &temp6x6b[0][0], 6, spajac_world, n, full_result, n);
      /* the enabled rows (mod.cpp:1484-1491) */
      ki = 0;
      for (tsri=0; tsri<6; tsri++) if (contsr->tsr_enabled[tsri])
      {
         memcpy(con_jacobian + ki*n, full_result + (tsri<3?tsri:8-tsri)*n, (size_t) n*sizeof(double));
         ki++;
      }
      free(spajac_world); free(full_result);
   }
   return 0;
}

/* `con_tsr all ...` / `everyn_tsr`: the mask and dimension (mod.cpp:2466-2480,2502-2518), one
 * constraint per moving point (mod.cpp:2582-2612) */
int ora_run_add_contsr(ora_run * r, int ee_link, const double tool[7], const double T0w[7], const double Twe[7], const double * Bw)
{
   run_contsr * ct = (run_contsr *) calloc(1, sizeof(run_contsr));
   int i;
   if (!ct) return -1;
   ct->r = r; ct->ee_link = ee_link;
   for (i=0; i<7; i++) { ct->tool[i] = tool[i]; ct->T0w[i] = T0w[i]; ct->Twe[i] = Twe[i]; }
   ct->k = 0;
   for (i=0; i<6; i++)
   {
      ct->Bw[i][0] = Bw[2*i]; ct->Bw[i][1] = Bw[2*i+1];
      if (ct->Bw[i][0] == 0.0 && ct->Bw[i][1] == 0.0) { ct->tsr_enabled[i] = 1; ct->k++; }
      else ct->tsr_enabled[i] = 0;
   }
   r->contsrs = (run_contsr **) realloc(r->contsrs, (size_t)(r->n_contsrs+1) * sizeof(run_contsr *));
   r->contsrs[r->n_contsrs++] = ct;
   for (i=0; i<r->c->m; i++)
      if (ora_chomp_add_constraint(r->c, ct->k, i, ct, con_tsr)) return -1;
   return ora_chomp_alloc_constraints(r->c);
}

int ora_run_eval_contsr(ora_run * r, int which, const double * point, double * h, double * J)
{
   double * pt;
   int i;
   if (which < 0 || which >= r->n_contsrs) return -1;
   pt = (double *) malloc((size_t) r->c->n * sizeof(double));
   for (i=0; i<r->c->n; i++) pt[i] = point[i];
   con_tsr(r->contsrs[which], r->c, 0, pt, h, J);
   free(pt);
   return r->contsrs[which]->k;
}

/* src/orcdchomp_mod.cpp:1800-2688 */
ora_run * ora_run_create(const ora_robot * rob, const double base_pose[7], const double * dofvals,
   int n_adof, const int * adofindices, const double * adofgoal, const double * basegoal,
   int n_sdfs, const ora_grid * const * grids, const double * poses_world_gsdf,
   const ora_run_params * params, const char ** errmsg)
{
   static const char * dummy;
   ora_run * r;
   int n, m, i, j, si, n_act, n_inact;
   int * is_active;
   if (!errmsg) errmsg = &dummy;
   *errmsg = 0;
   /* validation (mod.cpp:2091-2097) */
   if (!adofgoal) { *errmsg = "Did not pass either adofgoal or starttraj!"; return 0; }
   if (params->floating_base && !basegoal) { *errmsg = "Passed floating_base with no basegoal!"; return 0; }
   if (!params->floating_base && basegoal) { *errmsg = "Passed basegoal with no floating_base!"; return 0; }
   if (!n_sdfs) { *errmsg = "No signed distance fields have yet been computed!"; return 0; }
   if (params->lambda < 0.01) { *errmsg = "lambda must be >=0.01!"; return 0; }
   if (params->n_points < 3) { *errmsg = "n_points must be >=3!"; return 0; }
   if (params->floating_base && params->start_tsr) { *errmsg = "floating_base and start_tsr together is not yet implemented!"; return 0; }

   r = (ora_run *) calloc(1, sizeof(ora_run));
   r->robot = rob;
   r->n_points = params->n_points;
   r->floating_base = params->floating_base;
   r->epsilon = params->epsilon;
   r->epsilon_self = params->epsilon_self;
   r->obs_factor = params->obs_factor;
   r->obs_factor_self = params->obs_factor_self;
   r->use_hmc = params->use_hmc;
   r->hmc_resample_iter = 0;
   r->hmc_resample_lambda = params->hmc_resample_lambda;
   r->n_adof = n_adof;
   memcpy(r->base_pose, base_pose, 7*sizeof(double));
   r->dofvals = (double *) malloc((rob->n_dof ? rob->n_dof : 1) * sizeof(double));
   memcpy(r->dofvals, dofvals, rob->n_dof * sizeof(double));
   r->adofindices = (int *) malloc((n_adof ? n_adof : 1) * sizeof(int));
   memcpy(r->adofindices, adofindices, n_adof * sizeof(int));
   n = (r->floating_base ? 7 : 0) + n_adof;                       /* mod.cpp:2104-2105 */

   /* spheres of the robot and of every grabbed body (mod.cpp:2148-2300).  Body i = 0 is the robot, the others
    * follow in GetGrabbed() order (2168-2171).  A body without spheres is an error (2262-2263), also the robot. */
   {
      int n_eff = rob->n_spheres, gi, k, * eff_body;
      for (gi=0; gi<rob->n_grabbed; gi++) n_eff += rob->grabbed[gi].n_spheres;
      if (!rob->n_spheres) { run_free(r); *errmsg = "no spheres! kinbody does not have a <orcdchomp> tag defined?"; return 0; }
      for (gi=0; gi<rob->n_grabbed; gi++)
         if (!rob->grabbed[gi].n_spheres) { run_free(r); *errmsg = "no spheres! kinbody does not have a <orcdchomp> tag defined?"; return 0; }
      r->eff_link = (int *) malloc((size_t) n_eff * sizeof(int));
      r->eff_pos = (double *) malloc((size_t) n_eff * 3 * sizeof(double));
      r->eff_radius = (double *) malloc((size_t) n_eff * sizeof(double));
      eff_body = (int *) malloc((size_t) n_eff * sizeof(int));
      for (si=0; si<rob->n_spheres; si++)
      {
         r->eff_link[si] = rob->sphere_link[si];
         memcpy(r->eff_pos + 3*si, rob->sphere_pos + 3*si, 3*sizeof(double));
         r->eff_radius[si] = rob->sphere_radius[si];
         eff_body[si] = 0;
      }
      if (rob->n_grabbed)
      {
         /* T_w_rlink.inverse() * T_w_klink * pos with the transforms of the moment of create (2200-2208).  The link
          * frames of this model are 3x4 matrices (a base quaternion that is not of unit length scales them, as in
          * kin.c's expanded form), so the inverse is the matrix inverse */
         double * R = (double *) malloc((size_t) rob->n_links * 9 * sizeof(double));
         double * t = (double *) malloc((size_t) rob->n_links * 3 * sizeof(double));
         ora_robot_fk(rob, base_pose, dofvals, R, t, 0, 0);
         for (gi=0; gi<rob->n_grabbed; gi++)
         {
            const ora_grabbed * g = &rob->grabbed[gi];
            const double * A = R + 9*g->robot_link;
            double inv[9], det, Rk[3][3];
            inv[0] = A[4]*A[8] - A[5]*A[7]; inv[1] = A[2]*A[7] - A[1]*A[8]; inv[2] = A[1]*A[5] - A[2]*A[4];
            inv[3] = A[5]*A[6] - A[3]*A[8]; inv[4] = A[0]*A[8] - A[2]*A[6]; inv[5] = A[2]*A[3] - A[0]*A[5];
            inv[6] = A[3]*A[7] - A[4]*A[6]; inv[7] = A[1]*A[6] - A[0]*A[7]; inv[8] = A[0]*A[4] - A[1]*A[3];
            det = A[0]*inv[0] + A[1]*inv[3] + A[2]*inv[6];
            for (k=0; k<9; k++) inv[k] /= det;
            /* OpenRAVE's Transform * Vector rotates with the unit-quaternion matrix (geometry.h), the form this model's
             * link frames are made of (ora_robot_fk), not libcd's expanded one */
            ora_kin_quat_to_R(g->pose_world_klink + 3, Rk);
            for (k=0; k<g->n_spheres; k++, si++)
            {
               double pw[3], d[3];
               mat3_vec(&Rk[0][0], g->sphere_pos + 3*k, pw);
               for (j=0; j<3; j++) d[j] = pw[j] + g->pose_world_klink[j] - t[3*g->robot_link + j];
               mat3_vec(inv, d, r->eff_pos + 3*si);
               r->eff_link[si] = g->robot_link;
               r->eff_radius[si] = g->sphere_radius[k];
               eff_body[si] = 1 + gi;
            }
         }
         free(R); free(t);
      }

      /* which pairs of them a self-collision check would look at: the stand-in for OpenRAVE's CheckSelfCollision with grabbed
       * bodies (third party; src/orcdchomp_mod.cpp:2998-2999 calls it).  The robot's own spheres follow the link rule
       * (self_pairs over the ROBOT's spheres: adjacent links, links that touch with all dofs at zero); two spheres of one held
       * body are one rigid body; a held body is never tested against the link that holds it, nor against the links (by their own
       * spheres) and the other held bodies it overlapped AT THE MOMENT OF ITS GRAB -- OpenRAVE records what a body touches when
       * it is grabbed and ignores exactly that; what it comes to touch later counts.  For two held bodies the moment is the later
       * grab's (GetGrabbed() order is the order of the grabs). */
      {
         const int nl = rob->n_links;
         unsigned char * lex = (unsigned char *) malloc((size_t) nl * nl);
         double * R = (double *) malloc((size_t) nl * 9 * sizeof(double)), * t = (double *) malloc((size_t) nl * 3 * sizeof(double));
         double * pw = (double *) malloc((size_t) n_eff * 3 * sizeof(double));
         /* touch[g][x]: body 1 + g overlapped, at its grab, link x (x < nl) or body x - nl (x >= nl, an earlier body) */
         unsigned char * touch = (unsigned char *) calloc((size_t)(rob->n_grabbed + 1) * (nl + rob->n_grabbed + 1), 1);
         const int tw = nl + rob->n_grabbed + 1;
         int a, b2;
         self_pairs(rob, rob->n_spheres, rob->sphere_link, rob->sphere_pos, rob->sphere_radius, lex);
         for (gi=0; gi<rob->n_grabbed; gi++)
         {
            const ora_grabbed * g = &rob->grabbed[gi];
            /* everything where it was when body 1 + gi was grabbed: a held body's spheres ride on their link */
            ora_robot_fk(rob, g->has_grab_state ? g->grab_base_pose : base_pose, g->has_grab_state ? g->grab_dofvals : dofvals, R, t, 0, 0);
            for (a=0; a<n_eff; a++)
            {
               mat3_vec(R + 9*r->eff_link[a], r->eff_pos + 3*a, pw + 3*a);
               for (k=0; k<3; k++) pw[3*a+k] += t[3*r->eff_link[a] + k];
            }
            for (a=0; a<n_eff; a++)
            {
               if (eff_body[a] != 1 + gi) continue;
               for (b2=0; b2<n_eff; b2++)
               {
                  double d2 = 0.0;
                  if (eff_body[b2] > gi) continue;                /* itself, and bodies grabbed after it */
                  for (k=0; k<3; k++) { const double dd = pw[3*a+k] - pw[3*b2+k]; d2 += dd*dd; }
                  if (sqrt(d2) - (r->eff_radius[a] + r->eff_radius[b2]) < 0.0)
                     touch[gi*tw + (eff_body[b2] == 0 ? r->eff_link[b2] : nl + eff_body[b2])] = 1;
               }
            }
         }
         r->n_eff = n_eff;
         r->self_excl = (unsigned char *) calloc((size_t) n_eff * n_eff, 1);
         for (a=0; a<n_eff; a++)
         {
            r->self_excl[a*n_eff + a] = 1;
            for (b2=a+1; b2<n_eff; b2++)
            {
               const int ba = eff_body[a], bb = eff_body[b2];
               int ex;
               if (ba == 0 && bb == 0) ex = lex[r->eff_link[a]*nl + r->eff_link[b2]] != 0;
               else if (ba == bb) ex = 1;
               else if (ba == 0) ex = (r->eff_link[a] == r->eff_link[b2]) || touch[(bb-1)*tw + r->eff_link[a]];
               else if (bb == 0) ex = (r->eff_link[a] == r->eff_link[b2]) || touch[(ba-1)*tw + r->eff_link[b2]];
               else ex = (ba > bb) ? touch[(ba-1)*tw + nl + bb] : touch[(bb-1)*tw + nl + ba];
               r->self_excl[a*n_eff + b2] = r->self_excl[b2*n_eff + a] = (unsigned char) ex;
            }
         }
         free(lex); free(R); free(t); free(pw); free(touch);
      }

      /* active / inactive (2265-2291): is the sphere's robot link moved by an active dof */
      is_active = (int *) malloc((size_t) n_eff * sizeof(int));
      n_act = 0;
      for (si=0; si<n_eff; si++)
      {
         int act = r->floating_base ? 1 : 0;
         for (j=0; j<n_adof && !act; j++)
            if (ora_robot_does_affect(rob, adofindices[j], r->eff_link[si])) act = 1;
         is_active[si] = act;
         n_act += act;
      }
      n_inact = n_eff - n_act;
      if (!n_act) { free(is_active); free(eff_body); run_free(r); *errmsg = "robot active dofs must have at least one sphere!"; return 0; }
      r->n_spheres = n_eff;
      r->n_spheres_active = n_act;
      r->spheres = (run_sphere *) calloc((size_t) n_eff, sizeof(run_sphere));
      {
         int ia = 0, ii = n_act, body;
         /* The kdata list reverses a body's XML order and the head insertion in create reverses it again: a body's
          * spheres end in XML order.  Bodies are walked robot first and each body's spheres go to the HEAD of the
          * list, active and inactive alike (2273-2290; s_inactive_head is uninitialised there, treated as NULL):
          * the last grabbed body comes first, the robot last */
         for (body=rob->n_grabbed; body>=0; body--)
            for (si=0; si<n_eff; si++)
            {
               run_sphere * s;
               if (eff_body[si] != body) continue;
               s = is_active[si] ? &r->spheres[ia++] : &r->spheres[ii++];
               s->radius = r->eff_radius[si];
               s->robot_linkindex = r->eff_link[si];
               memcpy(s->pos_wrt_link, r->eff_pos + 3*si, 3*sizeof(double));
               s->xml_index = si;
            }
      }
      free(is_active); free(eff_body);
   }

   ora_rng_set(&r->rng, params->seed);                            /* mod.cpp:2303-2304 */

   m = r->n_points - 2;                                           /* mod.cpp:2315 */
   if (params->start_tsr) m++;
   r->J2 = (double *) malloc(3*n*sizeof(double));
   r->sphere_poss_all = (double *) calloc((size_t) r->n_points * n_act * 3, sizeof(double));
   r->sphere_poss = params->start_tsr ? r->sphere_poss_all : r->sphere_poss_all + n_act*3;   /* mod.cpp:2320-2323 */
   r->sphere_vels = (double *) calloc((size_t) m * n_act * 3, sizeof(double));
   r->sphere_accs = (double *) calloc((size_t) m * n_act * 3, sizeof(double));
   r->sphere_jacs = (double *) calloc((size_t) m * n_act * 3 * n, sizeof(double));
   r->fkR = (double *) malloc(rob->n_links * 9 * sizeof(double));
   r->fkt = (double *) malloc(rob->n_links * 3 * sizeof(double));
   r->fkaxis = (double *) malloc(rob->n_links * 3 * sizeof(double));
   r->fkanchor = (double *) malloc(rob->n_links * 3 * sizeof(double));

   /* world positions of the inactive spheres, once (mod.cpp:2332-2345) */
   r->sphere_poss_inactive = (double *) calloc((size_t)(n_inact ? n_inact : 1) * 3, sizeof(double));
   ora_robot_fk(rob, base_pose, dofvals, r->fkR, r->fkt, r->fkaxis, r->fkanchor);
   for (si=0; si<n_inact; si++)
   {
      const run_sphere * s = &r->spheres[n_act + si];
      double v[3];
      mat3_vec(r->fkR + 9*s->robot_linkindex, s->pos_wrt_link, v);
      for (j=0; j<3; j++) r->sphere_poss_inactive[si*3+j] = v[j] + r->fkt[3*s->robot_linkindex+j];
   }

   /* rooted sdfs (mod.cpp:2348-2369); the caller passes pose_world_gsdf already composed */
   r->n_rsdfs = n_sdfs;
   r->rsdfs = (ora_rsdf *) calloc(n_sdfs, sizeof(ora_rsdf));
   for (i=0; i<n_sdfs; i++)
   {
      r->rsdfs[i].grid = grids[i];
      memcpy(r->rsdfs[i].pose_world_gsdf, poses_world_gsdf + 7*i, 7*sizeof(double));
      ora_kin_pose_invert(r->rsdfs[i].pose_world_gsdf, r->rsdfs[i].pose_gsdf_world);
   }

   /* straight-line trajectory (mod.cpp:2417-2464) */
   r->traj = (double *) calloc((size_t) r->n_points * n, sizeof(double));
   if (r->floating_base)
   {
      for (j=0; j<7; j++) r->traj[j] = base_pose[j];
      for (j=0; j<n_adof; j++) r->traj[7+j] = dofvals[adofindices[j]];
      for (j=0; j<7; j++) r->traj[(r->n_points-1)*n+j] = basegoal[j];
      for (j=0; j<n_adof; j++) r->traj[(r->n_points-1)*n+7+j] = adofgoal[j];
   }
   else
   {
      for (j=0; j<n_adof; j++) r->traj[j] = dofvals[adofindices[j]];
      for (j=0; j<n_adof; j++) r->traj[(r->n_points-1)*n+j] = adofgoal[j];
   }
   /* note the in-place form: row 0 is rewritten first (to itself), the last row last */
   for (i=0; i<r->n_points; i++)
      for (j=0; j<n; j++)
         r->traj[i*n+j] = r->traj[j] + (r->traj[(r->n_points-1)*n+j]-r->traj[j]) * i/(r->n_points-1);
   if (r->floating_base)
      for (i=0; i<r->n_points; i++)
         ora_kin_pose_normalize(&r->traj[i*n]);

   /* the optimizer (mod.cpp:2521, 2567-2663) */
   if (ora_chomp_create(&r->c, m, n, params->D, &r->traj[(params->start_tsr?0:1)*n], n)) { run_free(r); *errmsg = "error creating chomp instance!"; return 0; }
   r->c->dt = 1.0/((r->n_points)-1);
   if (params->start_tsr)
   {
      /* mod.cpp:2570-2576: no start boundary in the metric, the start point held on the TSR */
      run_contsr * ct = (run_contsr *) calloc(1, sizeof(run_contsr));
      r->c->inits[0] = 0;
      ct->r = r; ct->ee_link = params->start_ee_link;
      for (i=0; i<7; i++) { ct->tool[i] = params->start_tool[i]; ct->T0w[i] = params->start_T0w[i]; ct->Twe[i] = params->start_Twe[i]; }
      for (i=0; i<6; i++)
      {
         ct->Bw[i][0] = params->start_Bw[2*i]; ct->Bw[i][1] = params->start_Bw[2*i+1];
         if (ct->Bw[i][0] == 0.0 && ct->Bw[i][1] == 0.0) { ct->tsr_enabled[i] = 1; ct->k++; }
      }
      r->start_contsr = ct;
      if (ora_chomp_add_constraint(r->c, ct->k, 0, ct, con_tsr) || ora_chomp_alloc_constraints(r->c))
         { run_free(r); *errmsg = "error adding the start_tsr constraint!"; return 0; }
   }
   else
      r->c->inits[0] = &r->traj[0*n];
   r->c->finals[0] = &r->traj[((r->n_points)-1)*n];
   r->c->cptr = r;
   r->c->cost_pre = sphere_cost_pre;
   r->c->cost = sphere_cost;
   r->c->lambda = params->lambda;
   if (params->use_momentum) r->c->use_momentum = 1;
   if (r->floating_base)
   {
      for (j=0; j<7; j++) { r->c->jlimit_lower[j] = -HUGE_VAL; r->c->jlimit_upper[j] = HUGE_VAL; }
      for (j=0; j<n_adof; j++)
      {
         r->c->jlimit_lower[7+j] = rob->limit_lower[adofindices[j]];
         r->c->jlimit_upper[7+j] = rob->limit_upper[adofindices[j]];
      }
   }
   else for (j=0; j<n_adof; j++)
   {
      r->c->jlimit_lower[j] = rob->limit_lower[adofindices[j]];
      r->c->jlimit_upper[j] = rob->limit_upper[adofindices[j]];
   }
   if (ora_chomp_init(r->c)) { run_free(r); *errmsg = "Error initializing chomp instance."; return 0; }
   return r;
}

/* src/orcdchomp_mod.cpp:2690-2852 */
static int run_iterate(ora_run * r, int n_iter, double * costs_out, double * trace, const double * noise, int n_noise)
{
   ora_chomp * c = r->c;
   double cost_total = 0.0, cost_obs = 0.0, cost_smooth = 0.0;
   int i, j, used_noise = 0;
   for (r->iter=0; r->iter<n_iter; r->iter++)
   {
      int ret;
      if (r->use_hmc && r->iter == r->hmc_resample_iter)           /* mod.cpp:2755-2768 */
      {
         double hmc_alpha = 100.0 * exp(0.02 * r->iter);
         if (noise)
         {
            if (used_noise < n_noise)
               memcpy(c->AG, noise + (size_t) used_noise * c->m * c->n, (size_t) c->m * c->n * sizeof(double));
            /* keep the rng stream aligned as if the gaussians had been drawn */
            for (i=0; i<c->m*c->n; i++) (void) ora_ran_gaussian(&r->rng, 1.0/sqrt(hmc_alpha));
            used_noise++;
         }
         else
            for (i=0; i<c->m; i++)
               for (j=0; j<c->n; j++)
                  c->AG[i*c->n+j] = ora_ran_gaussian(&r->rng, 1.0/sqrt(hmc_alpha));
         c->leapfrog_first = 1;
         r->hmc_resample_iter += 1 + (int) (- log(ora_rng_uniform(&r->rng)) / r->hmc_resample_lambda);
      }
      ret = ora_chomp_iterate(c, 1, &cost_total, &cost_obs, &cost_smooth);
      if (trace) { trace[3*r->iter+0] = cost_total; trace[3*r->iter+1] = cost_obs; trace[3*r->iter+2] = cost_smooth; }
      if (ret == -1) return -1;                                     /* mod.cpp:2799-2803 */
      if (r->floating_base)                                         /* mod.cpp:2806-2808 */
         for (i=0; i<r->n_points; i++)
            ora_kin_pose_normalize(&r->traj[i*c->n]);
   }
   ora_chomp_iterate(c, 0, &cost_total, &cost_obs, &cost_smooth);  /* mod.cpp:2830 */
   if (costs_out) { costs_out[0] = cost_total; costs_out[1] = cost_obs; costs_out[2] = cost_smooth; }
   return 0;
}

int ora_run_iterate(ora_run * r, int n_iter, double * costs_out, double * trace)
{
   return run_iterate(r, n_iter, costs_out, trace, 0, 0);
}

int ora_run_iterate_noise(ora_run * r, int n_iter, double * costs_out, double * trace, const double * noise, int n_noise)
{
   return run_iterate(r, n_iter, costs_out, trace, noise, n_noise);
}

void ora_run_destroy(ora_run * r) { run_free(r); }
int ora_run_n(const ora_run * r) { return r->c->n; }
int ora_run_m(const ora_run * r) { return r->c->m; }
int ora_run_n_points(const ora_run * r) { return r->n_points; }
int ora_run_n_spheres_active(const ora_run * r) { return r->n_spheres_active; }
int ora_run_n_spheres(const ora_run * r) { return r->n_spheres; }
double * ora_run_traj(ora_run * r) { return r->traj; }
ora_chomp * ora_run_chomp(ora_run * r) { return r->c; }
int ora_run_hmc_resample_iter(const ora_run * r) { return r->hmc_resample_iter; }
int ora_run_iter(const ora_run * r) { return r->iter; }
void ora_run_set_traj(ora_run * r, const double * traj) { memcpy(r->traj, traj, (size_t) r->n_points * r->c->n * sizeof(double)); }

/* ---------------------------------------------------------------- gettraj */
/* The re-check of mod::gettraj, src/orcdchomp_mod.cpp:2958-3006: total C-space length over the
 * active dofs of the waypoints (2968-2984), step_time = duration * 0.04 / total_dist (2986-2987),
 * for (time=0; time<duration; time+=step_time) Sample + collision query (2989-3003).
 * Third party and therefore stand-ins, the same ones the product states:
 *   - the timing (RetimeActiveDOFTrajectory with LinearTrajectoryRetimer, 2905-2911): every segment
 *     takes max_j |dq_j| / vmax_j;
 *   - Sample(): linear interpolation on the segment that holds `time`;
 *   - the collision query (CheckCollision / CheckSelfCollision, 2998-2999): an active sphere whose
 *     centre reads a field value below its radius (the optimizer's own model, first field in list
 *     order).  The reference's loop stops at the first colliding sample unless
 *     no_collision_exception: the first contact in (sample, XML sphere, field) order is reported.
 * Floating base: the base columns are interpolated like the joints and renormalised.
 * Returns the number of samples walked; *collides 0/1 and the contact's time / XML sphere / field /
 * depth (radius - value). */
void ora_robot_self_pairs(const ora_robot * rob, unsigned char * excl);

int ora_run_collision_recheck(ora_run * r, const double * vmax /* [n_adof] */, int * collides, double * time_out,
   int * sphere_out, int * field_out, double * depth_out)
{
   const int n = r->c->n, np = r->n_points, col0 = r->floating_base ? 7 : 0;
   double * tcum = (double *) malloc((size_t) np * sizeof(double));
   double * q = (double *) malloc((size_t) r->robot->n_dof * sizeof(double));
   double total_dist = 0.0, duration, step_time, time;
   unsigned char * excl = 0;
   int i, j, samples = 0, seg = 0;
   *collides = 0; *time_out = -1.0; *sphere_out = -1; *field_out = -1; *depth_out = 0.0;
   tcum[0] = 0.0;
   for (i=1; i<np; i++)
   {
      double dt = 0.0, d2 = 0.0;
      for (j=col0; j<n; j++)
      {
         const double d = r->traj[i*n+j] - r->traj[(i-1)*n+j];
         const double v = vmax[j-col0] > 0.0 ? vmax[j-col0] : 1.0;
         if (fabs(d) / v > dt) dt = fabs(d) / v;
      }
      tcum[i] = tcum[i-1] + dt;
      for (j=col0; j<n; j++) d2 += pow(r->traj[(i-1)*n+j] - r->traj[i*n+j], 2);
      total_dist += sqrt(d2);
   }
   duration = tcum[np-1];
   step_time = duration * 0.04 / total_dist;                      /* mod.cpp:2986-2987 */
   memcpy(q, r->dofvals, (size_t) r->robot->n_dof * sizeof(double));
   for (time=0.0; time<duration && !*collides; time+=step_time)
   {
      double u, base[7];
      int si, fi, k;
      while (seg < np-2 && tcum[seg+1] < time) seg++;
      u = (tcum[seg+1] > tcum[seg]) ? (time - tcum[seg]) / (tcum[seg+1] - tcum[seg]) : 0.0;
      memcpy(base, r->base_pose, sizeof(base));
      if (col0)
      {
         for (k=0; k<7; k++) base[k] = r->traj[seg*n+k] + (r->traj[(seg+1)*n+k] - r->traj[seg*n+k]) * u;
         ora_kin_pose_normalize(base);
      }
      for (j=0; j<r->n_adof; j++)
         q[r->adofindices[j]] = r->traj[seg*n+col0+j] + (r->traj[(seg+1)*n+col0+j] - r->traj[seg*n+col0+j]) * u;
      ora_robot_fk(r->robot, base, q, r->fkR, r->fkt, 0, 0);
      /* spheres in XML order: the robot's, then those of the bodies it held at create (the reference's note at
       * 2992-2996: CheckCollision and RobotBase::CheckSelfCollision include the grabbed bodies) */
      for (si=0; si<r->n_spheres && !*collides; si++)
      {
         double pw[3];
         int li = r->eff_link[si], active = r->floating_base;
         for (j=0; j<r->n_adof && !active; j++) active = ora_robot_does_affect(r->robot, r->adofindices[j], li);
         if (!active) continue;
         mat3_vec(r->fkR + 9*li, r->eff_pos + 3*si, pw);
         for (k=0; k<3; k++) pw[k] += r->fkt[3*li+k];
         for (fi=0; fi<r->n_rsdfs; fi++)
         {
            double pg[3], val;
            ora_kin_pose_compos(r->rsdfs[fi].pose_gsdf_world, pw, pg);
            if (ora_grid_double_interp(r->rsdfs[fi].grid, pg, &val)) continue;
            if (val - r->eff_radius[si] < 0.0)
            {
               *collides = 1; *time_out = time; *sphere_out = si; *field_out = fi;
               *depth_out = r->eff_radius[si] - val;
               break;
            }
         }
      }
      /* ... || boostrobot->CheckSelfCollision(report)  (src/orcdchomp_mod.cpp:2998-2999): the sphere model's
       * stand-in: two spheres on links that may collide (ora_robot_self_pairs) overlap; first pair in XML order */
      if (!*collides)
      {
         int a, b2;
         for (a=0; a<r->n_spheres && !*collides; a++)
            for (b2=a+1; b2<r->n_spheres; b2++)
            {
               const int la = r->eff_link[a], lb = r->eff_link[b2];
               double pa[3], pb[3], d2 = 0.0, rs, dist;
               if (r->self_excl[a * r->n_eff + b2]) continue;      /* (taken at create: run_self_pairs above) */
               mat3_vec(r->fkR + 9*la, r->eff_pos + 3*a, pa);
               mat3_vec(r->fkR + 9*lb, r->eff_pos + 3*b2, pb);
               for (k=0; k<3; k++) { const double d = (pa[k] + r->fkt[3*la+k]) - (pb[k] + r->fkt[3*lb+k]); d2 += d*d; }
               rs = r->eff_radius[a] + r->eff_radius[b2];
               dist = sqrt(d2);
               if (dist - rs < 0.0)
               {
                  *collides = 1; *time_out = time; *sphere_out = a; *field_out = -2 - b2;
                  *depth_out = rs - dist;
                  break;
               }
            }
      }
      samples++;
   }
   free(tcum); free(q); free(excl);
   return samples;
}

/* Which pairs of links a self-collision check looks at.  OpenRAVE's CheckSelfCollision (third party) skips
 * adjacent links: links joined by a joint, and links that already touch in the robot's initial configuration.
 * The sphere model's restatement: excl[la][lb] = 1 when la == lb, when one is the other's parent, when the robot
 * description declares the pair adjacent, or when any sphere of la overlaps any sphere of lb with all dofs at zero. */
static void self_pairs(const ora_robot * rob, int ns, const int * link, const double * pos, const double * radius,
   unsigned char * excl /* [n_links][n_links] */)
{
   const int nl = rob->n_links;
   const double ident[7] = { 0, 0, 0, 0, 0, 0, 1 };
   double * q = (double *) calloc((size_t)(rob->n_dof > 0 ? rob->n_dof : 1), sizeof(double));
   double * R = (double *) malloc((size_t) nl * 9 * sizeof(double)), * t = (double *) malloc((size_t) nl * 3 * sizeof(double));
   int a, b2, k;
   memset(excl, 0, (size_t) nl * nl);
   for (a=0; a<nl; a++)
   {
      excl[a*nl + a] = 1;
      if (rob->parent[a] >= 0) { excl[a*nl + rob->parent[a]] = 1; excl[rob->parent[a]*nl + a] = 1; }
   }
   for (a=0; a<rob->n_adjacent; a++)
   {
      excl[rob->adjacent[2*a]*nl + rob->adjacent[2*a+1]] = 1;
      excl[rob->adjacent[2*a+1]*nl + rob->adjacent[2*a]] = 1;
   }
   ora_robot_fk(rob, ident, q, R, t, 0, 0);
   for (a=0; a<ns; a++)
      for (b2=a+1; b2<ns; b2++)
      {
         const int la = link[a], lb = link[b2];
         double pa[3], pb[3], d2 = 0.0, rs = radius[a] + radius[b2];
         mat3_vec(R + 9*la, pos + 3*a, pa);
         mat3_vec(R + 9*lb, pos + 3*b2, pb);
         for (k=0; k<3; k++) { const double d = (pa[k] + t[3*la+k]) - (pb[k] + t[3*lb+k]); d2 += d*d; }
         if (sqrt(d2) - rs < 0.0) { excl[la*nl + lb] = 1; excl[lb*nl + la] = 1; }
      }
   free(q); free(R); free(t);
}

void ora_robot_self_pairs(const ora_robot * rob, unsigned char * excl /* [n_links][n_links] */)
{
   self_pairs(rob, rob->n_spheres, rob->sphere_link, rob->sphere_pos, rob->sphere_radius, excl);
}

/* create's starttraj branch, src/orcdchomp_mod.cpp:2375-2416 (fixed base): row i of the run's
 * trajectory is starttraj->Sample(i * duration / (n_points-1)) over the active dofs.  The document
 * is `count` waypoints with deltatimes (OpenRAVE's trajectory class is third party: linear
 * interpolation between the waypoints that bracket the time).  out [n_points][dof]. */
void ora_sample_starttraj(int count, int dof, const double * wp, const double * deltatime, int n_points, double * out)
{
   double * tcum = (double *) malloc((size_t) count * sizeof(double));
   double duration;
   int i, j, seg = 0;
   tcum[0] = 0.0;
   for (i=1; i<count; i++) tcum[i] = tcum[i-1] + deltatime[i];
   duration = tcum[count-1];
   for (i=0; i<n_points; i++)
   {
      const double t = i * duration / (n_points - 1);             /* mod.cpp:2412 */
      double u;
      int s0, s1;
      while (seg < count-2 && tcum[seg+1] < t) seg++;
      u = (count > 1 && tcum[seg+1] > tcum[seg]) ? (t - tcum[seg]) / (tcum[seg+1] - tcum[seg]) : 0.0;
      s0 = seg; s1 = (count > 1) ? seg+1 : seg;
      /* at (and past) the end a trajectory is its last waypoint (OpenRAVE's Sample), also when its last segments take no time */
      if (duration > 0.0 && (i == n_points-1 || t >= duration)) { s0 = s1 = count-1; u = 0.0; }
      for (j=0; j<dof; j++)
      {
         const double a0 = wp[s0*dof+j], a1 = wp[s1*dof+j];
         out[i*dof+j] = a0 + (a1 - a0) * u;
      }
   }
   free(tcum);
}

/* create's starttraj branch with floating_base, src/orcdchomp_mod.cpp:2378-2404: the base rows are sampled from
 * the document's `affine_transform` group (OpenRAVE order x y z qw qx qy qz, linear interpolation), reordered to
 * libcd's x y z qx qy qz qw and normalised (cd_kin_pose_normalize); the arm columns as above.
 * wp_joint [count][n_adof], wp_base [count][7]; out [n_points][7 + n_adof]. */
void ora_sample_starttraj_floating(int count, int n_adof, const double * wp_joint, const double * wp_base, const double * deltatime,
   int n_points, double * out)
{
   const int n = 7 + n_adof;
   double * tcum = (double *) malloc((size_t) count * sizeof(double));
   double duration;
   int i, j, seg = 0;
   tcum[0] = 0.0;
   for (i=1; i<count; i++) tcum[i] = tcum[i-1] + deltatime[i];
   duration = tcum[count-1];
   for (i=0; i<n_points; i++)
   {
      const double t = i * duration / (n_points - 1);             /* mod.cpp:2391, 2402 */
      double u, vec[7];
      int s0, s1;
      while (seg < count-2 && tcum[seg+1] < t) seg++;
      u = (count > 1 && tcum[seg+1] > tcum[seg]) ? (t - tcum[seg]) / (tcum[seg+1] - tcum[seg]) : 0.0;
      s0 = seg; s1 = (count > 1) ? seg+1 : seg;
      if (duration > 0.0 && (i == n_points-1 || t >= duration)) { s0 = s1 = count-1; u = 0.0; }      /* the end: the last waypoint */
      for (j=0; j<7; j++) vec[j] = wp_base[s0*7+j] + (wp_base[s1*7+j] - wp_base[s0*7+j]) * u;
      out[i*n+0] = vec[0]; out[i*n+1] = vec[1]; out[i*n+2] = vec[2];          /* mod.cpp:2392-2398 */
      out[i*n+3] = vec[4]; out[i*n+4] = vec[5]; out[i*n+5] = vec[6]; out[i*n+6] = vec[3];
      ora_kin_pose_normalize(&out[i*n]);
      for (j=0; j<n_adof; j++)
         out[i*n+7+j] = wp_joint[s0*n_adof+j] + (wp_joint[s1*n_adof+j] - wp_joint[s0*n_adof+j]) * u;
   }
   free(tcum);
}

/* gettraj's second trajectory for a floating base, src/orcdchomp_mod.cpp:2912-2949: per waypoint
 * [deltatime, affine_transform x y z qw qx qy qz, affine_velocities (the same order)], the velocities being the
 * differences to the previous waypoint over its deltatime (zero for the first).  traj [n_points][n] (libcd order
 * x y z qx qy qz qw in columns 0..6); out [n_points][15]. */
void ora_gettraj_affine_groups(const double * traj, int n_points, int n, const double * deltatime, double * out)
{
   int i;
   for (i=0; i<n_points; i++)
   {
      double * vec = out + (size_t) i*15;
      int k;
      for (k=0; k<15; k++) vec[k] = 0.0;
      vec[1+0] = traj[i*n+0]; vec[1+1] = traj[i*n+1]; vec[1+2] = traj[i*n+2];
      vec[1+4] = traj[i*n+3]; vec[1+5] = traj[i*n+4]; vec[1+6] = traj[i*n+5]; vec[1+3] = traj[i*n+6];
      if (i > 0)
      {
         const double dt = deltatime[i];
         vec[0] = dt;
         vec[1+7+0] = (traj[i*n+0] - traj[(i-1)*n+0]) / dt;
         vec[1+7+1] = (traj[i*n+1] - traj[(i-1)*n+1]) / dt;
         vec[1+7+2] = (traj[i*n+2] - traj[(i-1)*n+2]) / dt;
         vec[1+7+4] = (traj[i*n+3] - traj[(i-1)*n+3]) / dt;
         vec[1+7+5] = (traj[i*n+4] - traj[(i-1)*n+4]) / dt;
         vec[1+7+6] = (traj[i*n+5] - traj[(i-1)*n+5]) / dt;
         vec[1+7+3] = (traj[i*n+6] - traj[(i-1)*n+6]) / dt;
      }
   }
}

void ora_run_sphere_order(const ora_run * r, int * idx)
{
   int i;
   for (i=0; i<r->n_spheres; i++) idx[i] = r->spheres[i].xml_index;
}

int ora_run_eval_obstacle(ora_run * r, double * G, double * costs, double * sphere_poss_all)
{
   ora_chomp * c = r->c;
   int i;
   sphere_cost_pre(r, c, c->m, c->T_points);
   for (i=0; i<c->m; i++)
      sphere_cost(r, c, i, c->T_points[i], 0, costs ? &costs[i] : 0, G ? &G[i*c->n] : 0);
   if (sphere_poss_all)
      memcpy(sphere_poss_all, r->sphere_poss_all, (size_t) r->n_points * r->n_spheres_active * 3 * sizeof(double));
   return 0;
}

int ora_batch_run(const ora_robot * rob, const double base_pose[7], const double * dofvals,
   int n_adof, const int * adofindices, int n_runs, const double * adofgoals, const double * basegoals,
   int n_sdfs, const ora_grid * const * grids, const double * poses_world_gsdf,
   const ora_run_params * params, const unsigned int * seeds, int n_iter,
   double * traj_out, double * costs_out, int * status_out, int max_threads)
{
   int threads = 1, k;
   int n = (params->floating_base ? 7 : 0) + n_adof;
#ifdef _OPENMP
   threads = omp_get_max_threads();
   if (max_threads > 0 && max_threads < threads) threads = max_threads;
#pragma omp parallel for schedule(dynamic) num_threads(threads)
#else
   (void) max_threads;
#endif
   for (k=0; k<n_runs; k++)
   {
      ora_run_params p = *params;
      const char * err = 0;
      ora_run * r;
      double costs[3] = {0.0, 0.0, 0.0};
      int st;
      if (seeds) p.seed = seeds[k];
      r = ora_run_create(rob, base_pose, dofvals, n_adof, adofindices, adofgoals + (size_t) k*n_adof,
         basegoals ? basegoals + (size_t) k*7 : 0, n_sdfs, grids, poses_world_gsdf, &p, &err);
      if (!r) { if (status_out) status_out[k] = -2; continue; }
      st = ora_run_iterate(r, n_iter, costs, 0);
      if (traj_out) memcpy(traj_out + (size_t) k * p.n_points * n, r->traj, (size_t) p.n_points * n * sizeof(double));
      if (costs_out) memcpy(costs_out + 3*(size_t) k, costs, 3*sizeof(double));
      if (status_out) status_out[k] = st;
      ora_run_destroy(r);
   }
   return threads;
}

void ora_run_self_excluded(const ora_run * r, unsigned char * excl)
{
   memcpy(excl, r->self_excl, (size_t) r->n_eff * r->n_eff);
}
