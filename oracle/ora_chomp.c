/* ora_chomp.c -- TEST INFRASTRUCTURE (see oracle.h).
 * CPU restatement of libcd's cd_chomp optimizer core (src/libcd/chomp.c),
 * dense m x m algebra exactly as the reference structures it, with the
 * CBLAS/LAPACKE calls replaced by plain loops (third-party, absent here).
 * Hard constraints (chomp.c:219-234,405-425,553-600) are out of scope.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* C(MxN) = alpha * op(A)(MxK) * B(KxN) + beta * C, row-major.
 * stands in for cblas_dgemm(RowMajor, transA, NoTrans, ...) */
static void gemm(int transA, int M, int N, int K, double alpha,
   const double * A, int lda, const double * B, int ldb, double beta, double * C, int ldc)
{
   int i, j, k;
   for (i=0; i<M; i++)
   for (j=0; j<N; j++)
   {
      double s = 0.0;
      for (k=0; k<K; k++)
         s += (transA ? A[k*lda+i] : A[i*lda+k]) * B[k*ldb+j];
      C[i*ldc+j] = alpha*s + (beta == 0.0 ? 0.0 : beta*C[i*ldc+j]);
   }
}

/* in-place inverse by Gauss-Jordan with partial pivoting;
 * stands in for LAPACKE_dgetrf + LAPACKE_dgetri (chomp.c:393-403) */
static int invert(double * M, int n)
{
   double * aug = (double *) malloc((size_t) n * 2*n * sizeof(double));
   int i, j, k;
   if (!aug) return -1;
   for (i=0; i<n; i++)
   {
      for (j=0; j<n; j++) { aug[i*2*n+j] = M[i*n+j]; aug[i*2*n+n+j] = (i==j) ? 1.0 : 0.0; }
   }
   for (k=0; k<n; k++)
   {
      int piv = k;
      double best = fabs(aug[k*2*n+k]), d;
      for (i=k+1; i<n; i++) if (fabs(aug[i*2*n+k]) > best) { best = fabs(aug[i*2*n+k]); piv = i; }
      if (best == 0.0) { free(aug); return -2; }
      if (piv != k)
         for (j=0; j<2*n; j++) { double t = aug[k*2*n+j]; aug[k*2*n+j] = aug[piv*2*n+j]; aug[piv*2*n+j] = t; }
      d = 1.0 / aug[k*2*n+k];
      for (j=0; j<2*n; j++) aug[k*2*n+j] *= d;
      for (i=0; i<n; i++)
      {
         double f;
         if (i == k) continue;
         f = aug[i*2*n+k];
         if (f == 0.0) continue;
         for (j=0; j<2*n; j++) aug[i*2*n+j] -= f * aug[k*2*n+j];
      }
   }
   for (i=0; i<n; i++) for (j=0; j<n; j++) M[i*n+j] = aug[i*2*n+n+j];
   free(aug);
   return 0;
}

static double * zeros(size_t count)
{
   return (double *) calloc(count ? count : 1, sizeof(double));
}

/* src/libcd/chomp.c:40-178 */
int ora_chomp_create(ora_chomp ** cp, int m, int n, int D, double * T, int ldt)
{
   ora_chomp * c = (ora_chomp *) calloc(1, sizeof(ora_chomp));
   int i;
   if (!c) return -1;
   c->m = m; c->n = n; c->D = D;
   c->lambda = 1.0;
   c->dt = 1.0/(m+1);
   c->T = T; c->ldt = ldt;
   c->leapfrog_first = 1;
   c->T_points = (double **) malloc(m * sizeof(double *));
   c->G_points = (double **) malloc(m * sizeof(double *));
   c->AG_points = (double **) malloc(m * sizeof(double *));
   c->G = zeros((size_t) m*n);
   c->AG = zeros((size_t) m*n);                      /* zero momentum, chomp.c:114-115 */
   c->wds = zeros(D);
   c->initsfinals = zeros((size_t) 2*D*n);           /* non-NULL zero vectors, chomp.c:131-141 */
   c->inits = (double **) malloc((D ? D : 1) * sizeof(double *));
   c->finals = (double **) malloc((D ? D : 1) * sizeof(double *));
   c->A = zeros((size_t) m*m);
   c->Ainv = zeros((size_t) m*m);
   c->B = zeros((size_t) m*n);
   c->cost_nxn = zeros((size_t) n*n);
   c->cost_mxn = zeros((size_t) m*n);
   c->vels = zeros((size_t) m*n);
   c->jlimit_lower = zeros(n);
   c->jlimit_upper = zeros(n);
   c->Gjlimit = zeros((size_t) m*n);
   c->GjlimitAinv = zeros((size_t) m*n);
   c->Kvels = zeros((size_t) m*m);
   c->Evels = zeros((size_t) m*n);
   for (i=0; i<m; i++)
   {
      c->T_points[i] = &T[i*ldt];
      c->G_points[i] = &c->G[i*n];
      c->AG_points[i] = &c->AG[i*n];
   }
   for (i=0; i<D; i++)
   {
      c->wds[i] = (i<D-1) ? 0.0 : 1.0;               /* chomp.c:127-128 */
      c->inits[i] = &c->initsfinals[(2*i)*n];
      c->finals[i] = &c->initsfinals[(2*i+1)*n];
   }
   for (i=0; i<n; i++) { c->jlimit_lower[i] = -HUGE_VAL; c->jlimit_upper[i] = HUGE_VAL; }
   *cp = c;
   return 0;
}

void ora_chomp_free(ora_chomp * c)
{
   if (!c) return;
   free(c->T_points); free(c->G_points); free(c->AG_points);
   free(c->G); free(c->AG); free(c->wds); free(c->initsfinals);
   free(c->inits); free(c->finals); free(c->A); free(c->Ainv); free(c->B);
   free(c->cost_nxn); free(c->cost_mxn); free(c->vels);
   free(c->jlimit_lower); free(c->jlimit_upper); free(c->Gjlimit); free(c->GjlimitAinv);
   free(c->Kvels); free(c->Evels);
   free(c->cons_h); free(c->cons_Jcol); free(c->cons_JAJT); free(c->cons_ipiv); free(c->cons_delta);
   while (c->cons) { ora_chomp_con * con = c->cons; c->cons = con->next; free(con); }
   free(c);
}

/* src/libcd/chomp.c:219-234 (the list grows at its head) */
int ora_chomp_add_constraint(ora_chomp * c, int k, int i, void * cptr,
   int (*con_eval)(void * cptr, struct ora_chomp * c, int i, double * point, double * con_val, double * con_jacobian))
{
   ora_chomp_con * con = (ora_chomp_con *) malloc(sizeof(ora_chomp_con));
   if (!con) return -1;
   con->k = k; con->i = i; con->cptr = cptr; con->con_eval = con_eval;
   con->h = 0; con->J = 0;
   con->next = c->cons;
   c->cons = con;
   return 0;
}

/* src/libcd/chomp.c:405-425 (tail of cd_chomp_init) */
int ora_chomp_alloc_constraints(ora_chomp * c)
{
   ora_chomp_con * con;
   free(c->cons_h); free(c->cons_Jcol); free(c->cons_JAJT); free(c->cons_ipiv); free(c->cons_delta);
   c->cons_h = c->cons_Jcol = c->cons_JAJT = c->cons_delta = 0; c->cons_ipiv = 0;
   c->cons_k = 0;
   for (con=c->cons; con; con=con->next) c->cons_k += con->k;
   if (c->cons_k)
   {
      c->cons_h = (double *) malloc((size_t) c->cons_k * sizeof(double));
      c->cons_Jcol = (double *) malloc((size_t) c->cons_k * c->n * sizeof(double));
      c->cons_JAJT = (double *) malloc((size_t) c->cons_k * c->cons_k * sizeof(double));
      c->cons_ipiv = (int *) malloc((size_t) c->cons_k * sizeof(int));
      c->cons_delta = (double *) malloc((size_t) c->n * sizeof(double));
      if (!c->cons_h || !c->cons_Jcol || !c->cons_JAJT || !c->cons_ipiv || !c->cons_delta) return -1;
      c->cons_k = 0;
      for (con=c->cons; con; con=con->next)
      {
         con->h = c->cons_h + c->cons_k;
         con->J = c->cons_Jcol + (size_t) c->cons_k * c->n;
         c->cons_k += con->k;
      }
   }
   return 0;
}

/* A x = b for one right-hand side, LU with partial pivoting (rows swapped for the largest
 * magnitude of the column, first one on ties), then the two triangular solves: what
 * LAPACKE_dgesv(LAPACK_ROW_MAJOR, n, 1, ...) computes (chomp.c:579-581; LAPACK's blocked update
 * order differs in rounding only).  A [n][n] row-major is overwritten by its factors, b by x.
 * Returns the 1-based column of a zero pivot, 0 when there is none. */
static int dgesv_one(int n, double * A, int * ipiv, double * b)
{
   int i, j, k, info = 0;
   /* dgetrf: the factors and the row interchanges only; b is not touched yet */
   for (k=0; k<n; k++)
   {
      int p = k; double big = fabs(A[(size_t) k*n+k]);
      for (i=k+1; i<n; i++) if (fabs(A[(size_t) i*n+k]) > big) { big = fabs(A[(size_t) i*n+k]); p = i; }
      ipiv[k] = p;
      if (A[(size_t) p*n+k] == 0.0) { if (!info) info = k+1; continue; }
      if (p != k)
      {
         double t;
         for (j=0; j<n; j++) { t = A[(size_t) k*n+j]; A[(size_t) k*n+j] = A[(size_t) p*n+j]; A[(size_t) p*n+j] = t; }
      }
      for (i=k+1; i<n; i++)
      {
         const double l = A[(size_t) i*n+k] / A[(size_t) k*n+k];
         A[(size_t) i*n+k] = l;
         for (j=k+1; j<n; j++) A[(size_t) i*n+j] -= l * A[(size_t) k*n+j];
      }
   }
   /* info > 0: dgesv never calls dgetrs, the right-hand side stays as it was (chomp.c:582-586 then
    * pushes the ORIGINAL h back through A^-1 J^T) */
   if (info) return info;
   /* dgetrs: interchanges, forward and back substitution */
   for (k=0; k<n; k++)                /* dlaswp: all interchanges first (the stored multipliers carry the later ones) */
   {
      const int p = ipiv[k];
      if (p != k) { const double t = b[k]; b[k] = b[p]; b[p] = t; }
   }
   for (k=0; k<n; k++)
      for (i=k+1; i<n; i++) b[i] -= A[(size_t) i*n+k] * b[k];
   for (k=n-1; k>=0; k--)
   {
      double sum = b[k];
      for (j=k+1; j<n; j++) sum -= A[(size_t) k*n+j] * b[j];
      b[k] = sum / A[(size_t) k*n+k];
   }
   return 0;
}

/* (for the tests: the solver above on its own) */
int ora_dgesv_one(int n, double * A, int * ipiv, double * b) { return dgesv_one(n, A, ipiv, b); }

/* src/libcd/chomp.c:239-340: A = sum_d wds[d]/N_d K_d^T K_d etc. */
static int add_KEs(ora_chomp * c)
{
   int D = c->D, m = c->m, n = c->n, d, i;
   int * nd = (int *) malloc((D+1) * sizeof(int));     /* nd[d+1] = rows of K_d; nd[0] = m */
   double ** Ks = (double **) calloc(D ? D : 1, sizeof(double *));
   double ** Es = (double **) calloc(D ? D : 1, sizeof(double *));
   nd[0] = m;
   for (d=0; d<D; d++)
   {
      int has_i = c->inits[d] ? 1 : 0, has_f = c->finals[d] ? 1 : 0;
      int prev = nd[d], rows = prev - 1 + has_i + has_f;
      double * diff = zeros((size_t) rows * prev);
      nd[d+1] = rows;
      Ks[d] = zeros((size_t) rows * m);
      Es[d] = zeros((size_t) rows * n);
      if (has_i)
      {
         diff[0] = 1.0/c->dt;
         for (i=0; i<n; i++) Es[d][i] += (-1.0/c->dt) * c->inits[d][i];
      }
      for (i=0; i<prev-1; i++)
      {
         diff[(has_i+i)*prev + i]   = -1.0/c->dt;
         diff[(has_i+i)*prev + i+1] =  1.0/c->dt;
      }
      if (has_f)
      {
         diff[(rows-1)*prev + (prev-1)] = -1.0/c->dt;
         for (i=0; i<n; i++) Es[d][(rows-1)*n + i] += (1.0/c->dt) * c->finals[d][i];
      }
      if (d == 0)
         memcpy(Ks[0], diff, (size_t) rows * prev * sizeof(double));
      else
      {
         gemm(0, rows, m, prev, 1.0, diff, prev, Ks[d-1], m, 0.0, Ks[d], m);
         gemm(0, rows, n, prev, 1.0, diff, prev, Es[d-1], n, 1.0, Es[d], n);
      }
      free(diff);
   }
   memset(c->A, 0, (size_t) m*m*sizeof(double));
   memset(c->B, 0, (size_t) m*n*sizeof(double));
   memset(c->cost_nxn, 0, (size_t) n*n*sizeof(double));
   for (d=0; d<D; d++)
   {
      double w = c->wds[d] / nd[d+1];
      gemm(1, m, m, nd[d+1], w, Ks[d], m, Ks[d], m, 1.0, c->A, m);
      gemm(1, m, n, nd[d+1], w, Ks[d], m, Es[d], n, 1.0, c->B, n);
      gemm(1, n, n, nd[d+1], w, Es[d], n, Es[d], n, 1.0, c->cost_nxn, n);
   }
   c->trC = 0.0;
   for (i=0; i<n; i++) c->trC += c->cost_nxn[i*n+i];
   c->trC *= 0.5;
   for (d=0; d<D; d++) { free(Ks[d]); free(Es[d]); }
   free(Ks); free(Es); free(nd);
   return 0;
}

/* src/libcd/chomp.c:342-428 */
int ora_chomp_init(ora_chomp * c)
{
   int m = c->m, n = c->n, i, j;
   memset(c->Kvels, 0, (size_t) m*m*sizeof(double));
   memset(c->Evels, 0, (size_t) m*n*sizeof(double));
   for (i=0; i<m; i++)
   {
      if (i == 0)
      {
         if (c->inits[0])
         {
            c->Kvels[0*m+1] = 0.5 / c->dt;
            for (j=0; j<n; j++) c->Evels[j] = c->inits[0][j] * (-0.5/c->dt);
         }
         else { c->Kvels[0*m+1] = 1.0/c->dt; c->Kvels[0*m+0] = -1.0/c->dt; }
      }
      else if (i < m-1)
      {
         c->Kvels[i*m+i+1] =  0.5/c->dt;
         c->Kvels[i*m+i-1] = -0.5/c->dt;
      }
      else
      {
         if (c->finals[0])
         {
            for (j=0; j<n; j++) c->Evels[i*n+j] = c->finals[0][j] * (0.5/c->dt);
            c->Kvels[i*m+i-1] = -0.5/c->dt;
         }
         else { c->Kvels[i*m+i] = 1.0/c->dt; c->Kvels[i*m+i-1] = -1.0/c->dt; }
      }
   }
   if (add_KEs(c)) return -1;
   memcpy(c->Ainv, c->A, (size_t) m*m*sizeof(double));
   if (invert(c->Ainv, m)) return -2;
   return 0;
}

/* src/libcd/chomp.c:430-683 */
int ora_chomp_iterate(ora_chomp * c, int do_iteration, double * costp_total, double * costp_obs, double * costp_smooth)
{
   int m = c->m, n = c->n, i, j;
   double cost_point = 0.0, cost_obs = 0.0, cost_smooth = 0.0;
   int want_cost = (costp_total || costp_obs) ? 1 : 0;
   int num_limadjs = 0;

   /* vels = Evels + Kvels*T  (chomp.c:449-451; unused by sphere_cost, kept for fidelity) */
   memcpy(c->vels, c->Evels, (size_t) m*n*sizeof(double));
   gemm(0, m, n, m, 1.0, c->Kvels, m, c->T, c->ldt, 1.0, c->vels, n);

   if (c->cost_pre) c->cost_pre(c->cptr, c, m, c->T_points);          /* chomp.c:463-464 */

   if (do_iteration) memset(c->G, 0, (size_t) m*n*sizeof(double));     /* chomp.c:474 */
   if (c->cost) for (i=0; i<m; i++)
   {
      c->cost(c->cptr, c, i, c->T_points[i], &c->vels[i*n],
         want_cost ? &cost_point : 0, do_iteration ? c->G_points[i] : 0);
      if (want_cost) cost_obs += cost_point;
   }
   if (want_cost) cost_obs /= m;                                       /* chomp.c:490-491 */
   for (i=0; i<m*n; i++) c->G[i] *= 1.0/m;                             /* chomp.c:492 */

   if (do_iteration)
   {
      /* G += A T + B  (chomp.c:515-522) */
      gemm(0, m, n, m, 1.0, c->A, m, c->T, c->ldt, 1.0, c->G, n);
      for (i=0; i<m*n; i++) c->G[i] += c->B[i];

      /* AG = Ainv G, or momentum accumulate (chomp.c:525-548) */
      if (!c->use_momentum)
         gemm(0, m, n, m, 1.0, c->Ainv, m, c->G, n, 0.0, c->AG, n);
      else if (c->leapfrog_first)
      {
         gemm(0, m, n, m, 0.5/c->lambda, c->Ainv, m, c->G, n, 1.0, c->AG, n);
         c->leapfrog_first = 0;
      }
      else
         gemm(0, m, n, m, 1.0/c->lambda, c->Ainv, m, c->G, n, 1.0, c->AG, n);

      /* hard constraints (chomp.c:550-600): takes the unconstrained update AG and moves the trajectory itself */
      if (c->cons_k)
      {
         ora_chomp_con * con1, * con2;
         int a, b2, q, err;
         /* each point constraint into h and J */
         for (con1=c->cons; con1; con1=con1->next)
            con1->con_eval(con1->cptr, c, con1->i, c->T_points[con1->i], con1->h, con1->J);
         /* h += -1/lambda J AG_i */
         for (con1=c->cons; con1; con1=con1->next)
            for (a=0; a<con1->k; a++)
            {
               double sum = 0.0;
               for (q=0; q<n; q++) sum += con1->J[a*n+q] * c->AG_points[con1->i][q];
               con1->h[a] += (-1.0/c->lambda) * sum;
            }
         /* J Ainv J^T */
         for (con1=c->cons; con1; con1=con1->next)
         for (con2=c->cons; con2; con2=con2->next)
         {
            const double ainv = c->Ainv[con1->i * m + con2->i];
            double * blk = &c->cons_JAJT[(size_t)((con1->h)-(c->cons_h))*c->cons_k + ((con2->h)-(c->cons_h))];
            for (a=0; a<con1->k; a++)
            for (b2=0; b2<con2->k; b2++)
            {
               double sum = 0.0;
               for (q=0; q<n; q++) sum += con1->J[a*n+q] * con2->J[b2*n+q];
               blk[(size_t) a*c->cons_k + b2] = ainv * sum;
            }
         }
         err = dgesv_one(c->cons_k, c->cons_JAJT, c->cons_ipiv, c->cons_h);
         if (err) c->cons_error = err;       /* "constraint inversion error!" and on it goes */
         /* back through Ainv to the trajectory */
         for (con1=c->cons; con1; con1=con1->next)
         {
            for (q=0; q<n; q++)
            {
               double sum = 0.0;
               for (a=0; a<con1->k; a++) sum += con1->J[a*n+q] * con1->h[a];
               c->cons_delta[q] = sum;
            }
            for (i=0; i<m; i++)
               for (q=0; q<n; q++)
                  c->T[i*c->ldt+q] += -1.0 * c->Ainv[i*m + con1->i] * c->cons_delta[q];
         }
      }

      /* T -= AG/lambda  (chomp.c:604-605) */
      for (i=0; i<m; i++)
         for (j=0; j<n; j++)
            c->T[i*c->ldt+j] += (-1.0/c->lambda) * c->AG[i*n+j];

      /* joint-limit projection (chomp.c:608-655) */
      for (num_limadjs=0; num_limadjs<1000; num_limadjs++)
      {
         double largest = 0.0, scale;
         size_t largest_idx = 0;
         memset(c->Gjlimit, 0, (size_t) m*n*sizeof(double));
         for (i=0; i<m; i++)
         for (j=0; j<n; j++)
         {
            double t = c->T_points[i][j];
            if (t < c->jlimit_lower[j])
            {
               c->Gjlimit[i*n+j] = c->jlimit_lower[j] - t;
               if (fabs(c->Gjlimit[i*n+j]) > largest) { largest = fabs(c->Gjlimit[i*n+j]); largest_idx = (size_t)(i*n+j); }
            }
            if (t > c->jlimit_upper[j])
            {
               c->Gjlimit[i*n+j] = c->jlimit_upper[j] - t;
               if (fabs(c->Gjlimit[i*n+j]) > largest) { largest = fabs(c->Gjlimit[i*n+j]); largest_idx = (size_t)(i*n+j); }
            }
         }
         if (largest == 0.0) break;
         gemm(0, m, n, m, 1.0, c->Ainv, m, c->Gjlimit, n, 0.0, c->GjlimitAinv, n);
         scale = 1.01 * c->Gjlimit[largest_idx] / c->GjlimitAinv[largest_idx];
         /* the reference daxpy runs over m*n contiguous doubles of T (assumes ldt==n) */
         for (i=0; i<m*n; i++) c->T[i] += scale * c->GjlimitAinv[i];
      }
      c->last_num_limadjs = num_limadjs;
      if (!(num_limadjs < 1000)) return -1;
   }

   /* smoothness cost on the updated T (chomp.c:660-677) */
   if (costp_total || costp_smooth)
   {
      gemm(0, m, n, m, 1.0, c->A, m, c->T, c->ldt, 0.0, c->cost_mxn, n);
      gemm(1, n, n, m, 0.5, c->T, c->ldt, c->cost_mxn, n, 0.0, c->cost_nxn, n);
      gemm(1, n, n, m, 1.0, c->B, n, c->T, c->ldt, 1.0, c->cost_nxn, n);
      for (i=0; i<n; i++) cost_smooth += c->cost_nxn[i*n+i];
      cost_smooth += c->trC;
   }
   if (costp_total) *costp_total = cost_obs + cost_smooth;
   if (costp_obs) *costp_obs = cost_obs;
   if (costp_smooth) *costp_smooth = cost_smooth;
   return 0;
}
