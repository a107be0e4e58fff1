// vox_tri.h -- cube against triangle, for the occupancy sweep of computedistancefield over a kinbody given as a triangle mesh
// (/root/reference src/orcdchomp_mod.cpp:462-531: `CheckCollision(cube)` against whatever geometry the kinbody has; the
// reference's own scene is meshes, scripts/test_wam7.py:23-28).  OpenRAVE's collision checker is third party: this is the
// separating-axis test of a box and a triangle (13 axes: the cube's three, the triangle's normal, the nine cross products of
// a cube axis with a triangle edge).  A mesh is a SURFACE, as it is for the reference's checker: a cube inside a closed mesh
// meets no triangle, and the flood fill that follows (src/orcdchomp_mod.cpp:540-548) makes what it cannot reach an obstacle.
// So that a closed mesh always gives a closed shell of cells, touching counts: the cube and the triangle are apart only when
// some axis separates them by MORE than `tol` (the box test of host_math.cpp wants an overlap of more than tol; a box whose face
// lies exactly on a cell boundary is one layer of cells fatter as a mesh than as a box).
// One definition for the host path (host_math.cpp) and the device kernel (sdf_kernels.hip): the same operations in the same
// order, no contraction into fused multiply-adds in either file, so the two agree bit for bit.
#pragma once

#if defined(__HIPCC__)
#define ORC_HD __host__ __device__
#else
#define ORC_HD
#endif

// cR [9] row major, ct [3]: the cube's frame in the world (x_world = cR x_cube + ct); h: its half-extent; v [9]: the
// triangle's three vertices in the world
ORC_HD inline bool orc_cube_tri_touch(const double * cR, const double * ct, double h, const double * v, double tol)
{
   // the triangle in the cube's frame
   double p[3][3];
   for (int q=0; q<3; q++)
   {
      const double d[3] = { v[3*q+0] - ct[0], v[3*q+1] - ct[1], v[3*q+2] - ct[2] };
      for (int i=0; i<3; i++) p[q][i] = d[0]*cR[0*3+i] + d[1]*cR[1*3+i] + d[2]*cR[2*3+i];
   }
   const double lim = h + tol;
   // the cube's axes
   for (int i=0; i<3; i++)
   {
      double lo = p[0][i], hi = p[0][i];
      if (p[1][i] < lo) lo = p[1][i];
      if (p[1][i] > hi) hi = p[1][i];
      if (p[2][i] < lo) lo = p[2][i];
      if (p[2][i] > hi) hi = p[2][i];
      if (lo > lim || hi < -lim) return false;
   }
   const double e[3][3] = { { p[1][0]-p[0][0], p[1][1]-p[0][1], p[1][2]-p[0][2] },
                            { p[2][0]-p[1][0], p[2][1]-p[1][1], p[2][2]-p[1][2] },
                            { p[0][0]-p[2][0], p[0][1]-p[2][1], p[0][2]-p[2][2] } };
   // the triangle's plane: |n . p0| against the cube's radius along n
   {
      const double n[3] = { e[0][1]*e[1][2] - e[0][2]*e[1][1], e[0][2]*e[1][0] - e[0][0]*e[1][2], e[0][0]*e[1][1] - e[0][1]*e[1][0] };
      const double len = sqrt(n[0]*n[0] + n[1]*n[1] + n[2]*n[2]);
      // (a triangle without an area -- three points on a line, to rounding -- has no plane: the other twelve axes decide, as for a segment)
      const double e00 = e[0][0]*e[0][0] + e[0][1]*e[0][1] + e[0][2]*e[0][2], e11 = e[1][0]*e[1][0] + e[1][1]*e[1][1] + e[1][2]*e[1][2];
      if (len > 1e-12 * sqrt(e00 * e11))
      {
         const double dist = (n[0]*p[0][0] + n[1]*p[0][1] + n[2]*p[0][2]) / len;
         const double r = h * ((fabs(n[0]) + fabs(n[1])) + fabs(n[2])) / len;
         if (fabs(dist) > r + tol) return false;
      }
   }
   // cube axis i x edge j
   for (int i=0; i<3; i++)
   {
      const int i1 = (i+1)%3, i2 = (i+2)%3;
      for (int j=0; j<3; j++)
      {
         // a = unit_i x e_j: components (i1) -e_j[i2], (i2) e_j[i1]
         const double a1 = -e[j][i2], a2 = e[j][i1];
         const double len = sqrt(a1*a1 + a2*a2);
         if (!(len > 0.0)) continue;                       // the edge is parallel to the axis: no new direction
         const double s0 = a1*p[0][i1] + a2*p[0][i2], s1 = a1*p[1][i1] + a2*p[1][i2], s2 = a1*p[2][i1] + a2*p[2][i2];
         double lo = s0, hi = s0;
         if (s1 < lo) lo = s1;
         if (s1 > hi) hi = s1;
         if (s2 < lo) lo = s2;
         if (s2 > hi) hi = s2;
         const double r = h * (fabs(a1) + fabs(a2));
         if (lo > r + tol*len || hi < -(r + tol*len)) return false;
      }
   }
   return true;
}
