/* ora_grid.c -- TEST INFRASTRUCTURE (see oracle.h).
 * CPU restatement of the 3-d double-cell paths of libcd's grid module.
 * Arithmetic order follows the reference expression by expression so that the
 * results are bit-identical to oracle/_ref (checked in tests/test_oracle_ref.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* src/libcd/grid.c:61-97 (create_sizeown), fixed to n=3 and double cells */
ora_grid * ora_grid_create(const int sizes[3], const double lengths[3], double init)
{
   ora_grid * g = (ora_grid *) calloc(1, sizeof(ora_grid));
   size_t i;
   if (!g) return 0;
   g->n = 3;
   g->ncells = 1;
   for (i=0; i<3; i++)
   {
      g->sizes[i] = sizes[i];
      g->lengths[i] = lengths ? lengths[i] : 1.0;
      g->ncells *= (size_t) sizes[i];
   }
   g->data = (double *) malloc(g->ncells * sizeof(double));
   if (!g->data) { free(g); return 0; }
   for (i=0; i<g->ncells; i++) g->data[i] = init;
   return g;
}

/* src/libcd/grid.c:99-132 */
ora_grid * ora_grid_copy(const ora_grid * src)
{
   ora_grid * g = ora_grid_create(src->sizes, src->lengths, 0.0);
   if (!g) return 0;
   memcpy(g->data, src->data, src->ncells * sizeof(double));
   return g;
}

void ora_grid_free(ora_grid * g)
{
   if (!g) return;
   free(g->data);
   free(g);
}

/* src/libcd/grid.c:145-158 */
int ora_grid_index_to_subs(const ora_grid * g, size_t index, int * subs)
{
   int d;
   for (d=g->n-1; d>=0; d--)
   {
      subs[d] = (int)(index % (size_t) g->sizes[d]);
      index /= (size_t) g->sizes[d];
   }
   return 0;
}

/* src/libcd/grid.c:172-189: centre = (0.5+sub)/size, then *= length */
int ora_grid_center_index(const ora_grid * g, size_t index, double * center)
{
   int d;
   for (d=g->n-1; d>=0; d--)
   {
      int sub = (int)(index % (size_t) g->sizes[d]);
      index /= (size_t) g->sizes[d];
      center[d] = (0.5 + sub) / g->sizes[d];
   }
   for (d=0; d<g->n; d++) center[d] *= g->lengths[d];
   return 0;
}

/* src/libcd/grid.c:191-209 */
int ora_grid_lookup_index(const ora_grid * g, const double * p, size_t * index)
{
   int d;
   size_t idx = 0;
   for (d=0; d<g->n; d++)
   {
      double x = p[d] / g->lengths[d];
      int sub;
      if (x < 0.0) return 1;
      if (x > 1.0) return 1;
      sub = (int) floor(x * g->sizes[d]);
      if (sub == g->sizes[d]) sub--;
      idx = idx * (size_t) g->sizes[d] + (size_t) sub;
   }
   *index = idx;
   return 0;
}

/* which neighbour the one-sided difference uses along one axis:
 * +1 = next cell, -1 = previous cell (src/libcd/grid.c:357-365, 418-424) */
static int pick_side(int sub, int size, double p, double center)
{
   if (sub == 0) return +1;
   if (sub == size-1) return -1;
   return (p < center) ? -1 : +1;
}

/* src/libcd/grid.c:331-384: no HUGE_VAL handling */
int ora_grid_double_grad(const ora_grid * g, const double * p, double * grad)
{
   size_t index, rem, stride;
   int d;
   if (ora_grid_lookup_index(g, p, &index)) return 1;
   stride = 1;
   rem = index;
   for (d=g->n-1; d>=0; d--)
   {
      int sub = (int)(rem % (size_t) g->sizes[d]);
      double center, diff;
      rem /= (size_t) g->sizes[d];
      center = (0.5 + sub) / g->sizes[d] * g->lengths[d];
      if (pick_side(sub, g->sizes[d], p[d], center) < 0)
      {
         diff = g->data[index];
         diff -= g->data[index - stride];
      }
      else
      {
         diff = g->data[index + stride];
         diff -= g->data[index];
      }
      grad[d] = diff * g->sizes[d] / g->lengths[d];
      stride *= (size_t) g->sizes[d];
   }
   return 0;
}

/* src/libcd/grid.c:386-454: first-order expansion about the
 * containing cell centre, one-sided slopes, HUGE_VAL poisons the result */
int ora_grid_double_interp(const ora_grid * g, const double * p, double * valuep)
{
   size_t index, rem, stride;
   double value;
   int d;
   if (ora_grid_lookup_index(g, p, &index)) return 1;
   value = g->data[index];
   if (value == HUGE_VAL) { *valuep = HUGE_VAL; return 0; }
   stride = 1;
   rem = index;
   for (d=g->n-1; d>=0; d--)
   {
      int sub = (int)(rem % (size_t) g->sizes[d]);
      double center, after, before, diff, slope;
      rem /= (size_t) g->sizes[d];
      center = (0.5 + sub) / g->sizes[d] * g->lengths[d];
      if (pick_side(sub, g->sizes[d], p[d], center) < 0)
      {
         after = g->data[index];
         before = g->data[index - stride];
      }
      else
      {
         after = g->data[index + stride];
         before = g->data[index];
      }
      if (after == HUGE_VAL || before == HUGE_VAL) { *valuep = HUGE_VAL; return 0; }
      diff = after;
      diff -= before;
      slope = diff * g->sizes[d] / g->lengths[d];
      value += slope * (p[d] - center);
      stride *= (size_t) g->sizes[d];
   }
   *valuep = value;
   return 0;
}

/* src/libcd/grid.c:269-329: 1-d squared distance transform by lower envelope of
 * parabolas (Felzenszwalb & Huttenlocher); HUGE_VAL samples carry no parabola */
static void sedt_1d(int n, const double * f, double * out, size_t ostride, int * v, double * z)
{
   int q, k = 0, i;
   for (q=0; q<n; q++)
   {
      double s;
      if (f[q] == HUGE_VAL) continue;
      if (k == 0)
      {
         k = 1; v[0] = q; z[0] = -HUGE_VAL; z[1] = HUGE_VAL;
         continue;
      }
      for (;;)
      {
         s = f[q] + q*q;
         s -= f[v[k-1]] + v[k-1]*v[k-1];
         s /= 2.0 * (q - v[k-1]);
         if (s <= z[k-1]) { k--; continue; }
         break;
      }
      k++;
      v[k-1] = q;
      z[k-1] = s;
      z[k] = HUGE_VAL;
   }
   if (k == 0)
   {
      for (i=0; i<n; i++) out[i*ostride] = HUGE_VAL;
      return;
   }
   k = 0;
   for (q=0; q<n; q++)
   {
      while (z[k+1] < q) k++;
      out[q*ostride] = pow(q - v[k], 2.0) + f[v[k]];
   }
}

/* src/libcd/grid.c:462-569: separable squared EDT, each axis scaled by (len/size)^2 */
int ora_grid_double_dt_sqeuc(ora_grid ** gp_dt, const ora_grid * g_func)
{
   ora_grid * g = ora_grid_copy(g_func);
   int axis;
   if (!g) return -1;
   for (axis=0; axis<3; axis++)
   {
      int dim_n = g->sizes[axis];
      size_t dim_stride = 1, outer, inner, no, ni;
      double res2 = pow(g_func->lengths[axis] / g->sizes[axis], 2.0);
      int * v = (int *) malloc(dim_n * sizeof(int));
      double * z = (double *) malloc((dim_n + 1) * sizeof(double));
      double * f = (double *) malloc(dim_n * sizeof(double));
      int a2, i;
      if (!v || !z || !f) { free(v); free(z); free(f); ora_grid_free(g); return -1; }
      for (a2=axis+1; a2<3; a2++) dim_stride *= (size_t) g->sizes[a2];
      no = 1; for (a2=0; a2<axis; a2++) no *= (size_t) g->sizes[a2];
      ni = dim_stride;
      /* every 1-d line along this axis (the reference walks them with a
       * carry-chain of subscripts; the set of lines is the same) */
      for (outer=0; outer<no; outer++)
      for (inner=0; inner<ni; inner++)
      {
         double * line = g->data + outer * (size_t) dim_n * dim_stride + inner;
         for (i=0; i<dim_n; i++) f[i] = line[i*dim_stride] / res2;
         sedt_1d(dim_n, f, line, dim_stride, v, z);
         for (i=0; i<dim_n; i++) line[i*dim_stride] *= res2;
      }
      free(v); free(z); free(f);
   }
   *gp_dt = g;
   return 0;
}

/* src/libcd/grid.c:637-687: input 0.0 = free, HUGE_VAL = obstacle;
 * output sqrt(dist^2 to obstacle) - sqrt(dist^2 to free): positive outside */
int ora_grid_double_bin_sdf(ora_grid ** gp_dt, const ora_grid * g_emp)
{
   ora_grid * g_obs = ora_grid_copy(g_emp);
   ora_grid * sedt_emp = 0, * sedt_obs = 0;
   size_t i;
   if (!g_obs) return -1;
   for (i=0; i<g_emp->ncells; i++)
      g_obs->data[i] = (g_emp->data[i] == 0.0) ? HUGE_VAL : 0.0;
   if (ora_grid_double_dt_sqeuc(&sedt_emp, g_emp)) { ora_grid_free(g_obs); return -1; }
   if (ora_grid_double_dt_sqeuc(&sedt_obs, g_obs)) { ora_grid_free(g_obs); ora_grid_free(sedt_emp); return -1; }
   for (i=0; i<sedt_obs->ncells; i++)
      sedt_obs->data[i] = sqrt(sedt_obs->data[i]) - sqrt(sedt_emp->data[i]);
   ora_grid_free(g_obs);
   ora_grid_free(sedt_emp);
   *gp_dt = sedt_obs;
   return 0;
}

/* src/libcd/grid_flood.c:30-111 with replace = replace_1_to_0
 * (src/orcdchomp_mod.cpp:160-168), no wrapping: axis neighbours only. */
long ora_grid_flood_fill_1_to_0(ora_grid * g, size_t index_start)
{
   size_t cap = 1024, top = 0;
   size_t * stack = (size_t *) malloc(cap * sizeof(size_t));
   long replaced = 0;
   if (!stack) return -1;
   stack[top++] = index_start;
   while (top)
   {
      size_t index = stack[--top];
      int subs[3], d, pm;
      if (g->data[index] != 1.0) continue;
      g->data[index] = 0.0;
      replaced++;
      ora_grid_index_to_subs(g, index, subs);
      for (d=0; d<3; d++)
      for (pm=0; pm<2; pm++)
      {
         int s = subs[d] + (pm==0 ? -1 : 1);
         size_t nidx;
         int save;
         if (s < 0 || s >= g->sizes[d]) continue;
         save = subs[d];
         subs[d] = s;
         nidx = ((size_t) subs[0] * g->sizes[1] + subs[1]) * g->sizes[2] + subs[2];
         subs[d] = save;
         if (top == cap)
         {
            size_t * ns;
            cap *= 2;
            ns = (size_t *) realloc(stack, cap * sizeof(size_t));
            if (!ns) { free(stack); return -1; }
            stack = ns;
         }
         stack[top++] = nidx;
      }
   }
   free(stack);
   return replaced;
}
