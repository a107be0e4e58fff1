// dev_types.h -- plain-old-data handed from the host layer to the HIP kernels.
//
// Naming follows the reference's domain (runs, waypoints, spheres, rooted sdfs):
//   struct run        /root/reference src/orcdchomp_mod.cpp:887-966
//   struct run_rsdf   src/orcdchomp_mod.cpp:850-855
//   struct cd_chomp   src/libcd/chomp.h:38-101
#pragma once

#define ORC_MAX_JOINTS   32      // active (optimized) joints of one robot
#define ORC_MAX_SPHERES  64      // spheres of one robot (active + inactive)
#define ORC_MAX_POINTS   4096    // moving waypoints of one run (the dense m x m matrices of the metric are built on the host at create)
#define ORC_MAX_SAVE     4       // saved frames while walking a kinematic tree
#define ORC_MAX_SDFS     8
#define ORC_BLOCK        256     // threads per workgroup: one workgroup per run
#ifndef ORC_WGS_PER_CU
#define ORC_WGS_PER_CU    3       // resident workgroups per CU the kernels' register budget is sized for (launch bounds)
#endif
#ifndef ORC_WGS_PER_CU_FP32_MANY
#define ORC_WGS_PER_CU_FP32_MANY 4   // ... of the fp32 kernels for more than 16 active spheres (128 VGPRs)
#endif
#define ORC_SS_MAX_RANK   4       // derivatives whose band inverse is applied through its generators (host_math.h has the same)
#define ORC_SCAN_RPL      4       // rows per lane of the scan solve: m <= 64*ORC_SCAN_RPL
#define ORC_VERDICT_NONE  0x7fffffffffffffffull      // key of a run without a contact (collision_verdict_kernel)
#define ORC_LDS_HEADER    256     // bytes in front of the LDS carve-up: reduction scratch [16] doubles, [8] ints, column masks, timer mark, phase counters [8]
#define ORC_LIM_LIST      64      // violated entries the sparse joint-limit rounds handle
#define ORC_LIM_SCRATCH  (256 + ORC_LIM_LIST*16)   // bytes: 4 wave records + header, entry list
#define ORC_PAIR_ROUNDS   16      // rounds of the dense self-collision pair list at most (cost_pairs.h): 16 x 32 lanes hold every pair of 32 spheres
#define ORC_PAIR_DEG      4       // pairs of one round that add to (and that subtract from) one sphere at most

// one optimized joint, in topological order.  Non-optimized joints are folded
// into the fixed transforms on the host when the batch is created.
template <typename real>
struct DevJoint
{
   real Rfix[9];      // rotation   from-frame -> joint frame (row major)
   real tfix[3];      // translation from-frame -> joint frame
   real axis[3];      // unit axis in the joint frame
   int type;          // 1 revolute, 2 prismatic
   int col;           // column of the trajectory / gradient this joint owns
   int load_slot;     // -2 base frame, -1 previous joint's moved frame, k>=0 saved slot k
   int save_slot;     // -1 none, k>=0: keep the moved frame in slot k
   int sph_begin;     // active spheres [sph_begin, sph_end) ride on this joint's moved frame
   int sph_end;
   int rfix_identity; // Rfix == I (skips a 3x3 product)
   int axis_kind;     // 1/2/3: the axis is +-x/+-y/+-z of the joint frame (cheap column rotation), 0 general
   real axis_sign;    // +-1 for axis_kind != 0
   int aff_begin;     // the active spheres this joint moves are [aff_begin, aff_end) when DevModel::jt_scan != 0
   int aff_end;
   int packed;        // the walk's control word in one LDS read: type | axis_kind<<2 | rfix_identity<<4 | (axis_sign<0)<<5 | sph_begin<<8 | sph_end<<16 | col<<24
   int packed2;       // (load_slot + 2) | (save_slot + 2) << 4: the tree's frame traffic, scalar like `packed`
};

// What the FK walk reads of one joint (fk.h), as one record: a burst of scalar loads per joint (32 words in fp32),
// the spheres riding on the joint's link included (their table entries used to be loaded one sphere at a time,
// each load a scalar-cache round trip in the middle of the joint's step)
template <typename real>
struct DevFkJoint
{
   real Rfix[9], tfix[3], axis[3];
   int ctl;             // spheres on the link: count (bits 0-7), index of the first in the sorted order (8-15); load_slot + 2 (16-19), save_slot + 2 (20-23); revolute (24); column (25-31)
   real sph[4][3];      // centres of the first four of them in the link frame (more: DevModel::sph_pos from the fifth on)
   int slot[4];         // their slots of the position buffer
};

template <typename real>
struct DevModel
{
   int nj;                 // optimized joints
   int n;                  // optimizer dofs = 7*floating + n_adof
   int floating;           // floating base: columns 0..6 are the base pose
   int tree;               // the joint tree branches (frames are saved/restored while walking it)
   int Sa;                 // lanes of the active block: active spheres, or 16 slots when the spheres are placed (see slot_of)
   int S;                  // Sa + inactive spheres
   int Sa_real;            // active spheres (sorted by the joint they ride on: the order of sph_pos, J^T ranges, slot_of)
   int placed;             // slot_of is not the identity
   unsigned long long live_mask;   // bit s: lane/slot s of the active block holds a sphere
   int slot_of[ORC_MAX_SPHERES];   // sorted index -> slot (lane of the DPP row / index of pos, sph_radius, sph_link, sph_affects); placed rows: entries Sa_real .. 15 name slots WITHOUT an active sphere (the J^T scan fetches an exact zero through them)
   int GS;                 // lanes per waypoint in the cost phase (power of two >= Sa)
   int base_sph_begin;     // active spheres fixed to the base frame (floating base only)
   int base_sph_end;
   int jt_scan;            // spheres moved by a joint are contiguous: 1 every set ends at Sa (chain), 2 general ranges, 0 not contiguous
   real base_R[9];         // base frame when not floating
   real base_t[3];
   DevJoint<real> joints[ORC_MAX_JOINTS];
   int jpacked[ORC_MAX_JOINTS];          // joints[j].packed / packed2 as arrays: the FK walk reads its control words
   int jpacked2[ORC_MAX_JOINTS];         // with scalar loads (no LDS round trip in front of every branch)
   real sph_pos[ORC_MAX_SPHERES][3];     // active, SORTED order: in the attach frame
   real sph_radius[ORC_MAX_SPHERES];
   int sph_link[ORC_MAX_SPHERES];        // robot link index (same-link test)
   unsigned long long sph_affects[ORC_MAX_SPHERES]; // bit j: joint j moves the sphere
   real sph_inactive_pos[ORC_MAX_SPHERES][3];       // world positions of inactive spheres [S-Sa]
   // Inactive spheres carried on free lanes of the 16-lane row (cost_gs16.h): they stand still, so their own
   // side of every pair is zero and the row rotations evaluate exactly the active sphere's side
   // (src/orcdchomp_mod.cpp:1262-1263, 1310-1311); they do not go through the loop over inactive spheres.
   int n_static;
   int static_slot[16];
   real static_pos[16][3];                          // world positions
   unsigned long long static_mask;                  // bit s: lane/slot s holds one of them
   DevFkJoint<real> fkj[ORC_MAX_JOINTS];            // the FK walk's records, depth-first order like `joints`
   // A robot that is a chain [0, fk_nanc) which then branches: the branches from joint fk_b_begin on can be walked by
   // another wavefront (which walks the chain as well, without storing): two walks of about half the length
   int fk_split, fk_nanc, fk_b_begin;
   // The dense self-collision pair list of the 32-lane family (cost_pairs.h; built at create: batch.cpp build_pair_table).
   // Every pair of lanes that can ever count (spheres on different links, at least one of them active) has ONE entry;
   // round r, lane k of a waypoint's lane group evaluates entry r*GS + k for that waypoint: the net force of the pair
   // on its FIRST sphere (the second receives the opposite).  Entries are in the order of how often the pair is within
   // range (fixed-seed configurations of the robot), so the rounds that are nearly always needed come first and are full.
   int pr_rounds;                                   // rounds in use (0: the robot does not use the list)
   int pr_ab[ORC_PAIR_ROUNDS*32];                   // first lane | second lane << 8; first == second: no pair
   int pr_gat[ORC_PAIR_ROUNDS*32*2];                // for lane s as a SPHERE in round r: the pair lanes (x4: ds_bpermute addresses within the group) whose force it adds (the four bytes of word 0) and subtracts (word 1); unused: the round's last lane, which never holds a pair
   int pr_hot;                                      // the first pr_hot rounds hold a pair that is (nearly) always within range: they are evaluated without asking
   real pr_rsum[ORC_PAIR_ROUNDS*32];                // r_first + r_second
};

// a rooted signed distance field (struct run_rsdf + struct cd_grid)
template <typename real>
struct DevSdf
{
   const real * data;      // C order [x][y][z]
   int size[3];
   int rot_identity;       // Rgw == Rwg == I exactly (the field is only translated)
   real length[3];
   real inv_length[3];     // 1/length
   real cell[3];           // length/size
   real size_over_len[3];  // size/length
   real Rgw[9];            // world -> grid: p_g = Rgw p + tgw   (pose_gsdf_world)
   real tgw[3];
   real Rwg[9];            // grid -> world rotation (pose_world_gsdf), for the gradient
};

// The same field for the many-sphere cost path (cost_generic.h), read with scalar loads: everything is
// expressed in units of cells, so that a lookup needs no quotient and no product with a cell size.
//   g = M p + t          grid coordinates of a world point, in cells (0 .. size)
//   value = v0 + sum_d (after_d - before_d) (g_d - (sub_d + 0.5))
//   world gradient = W (after - before)
// (struct cd_grid + run_rsdf of the reference; the formulae are those of src/libcd/grid.c:331-454 with the
// three quotients per axis folded into M, t and W on the host in double precision)
template <typename real>
struct DevSdfCell
{
   real M[9];              // diag(size/length) Rgw
   real t[3];              // diag(size/length) tgw
   real W[9];              // Rwg diag(size/length)
   real fsize[3];          // size as reals
   real fsize_m1[3];       // size - 1
   int stride_b[2];        // byte strides of the x and y axes (z: sizeof(real))
   const real * data;      // C order [x][y][z]
   real stride_r[3];       // the byte strides of the three axes as reals (fp64: cell offsets are formed by fused multiply-adds, exactly)
   real pad_;
};

// a TSR hard constraint on every moving point (struct run_contsr + struct tsr of the reference,
// src/orcdchomp_mod.cpp:873-885, src/orcdchomp_mod.h:80-87), folded onto the device's joint order
template <typename real>
struct DevTsr
{
   unsigned int chain_mask;   // bit j: device joint j moves the end effector's link
   int k;                     // rows = enabled entries of xyzrpy (Bw row == [0 0], mod.cpp:2466-2480)
   int enabled[6];
   int row_base;              // first row of this constraint's blocks in the system (list order: the last constraint first)
   int npts;                  // points the constraint holds: all m moving points (`con_tsr all`, `everyn_tsr`), or 1
   int point;                 // npts == 1: that moving point (`start_tsr`: 0)
   int blk_base;              // first block (constraint, point) of this constraint in list order
   real Xl_R[9], Xl_t[3];     // the link in the moved frame of its last chain joint (in the base frame when the chain is empty)
   real tool[7];              // end effector in the link frame
   real table_world[7];       // cd_kin_pose_invert(T0w)
   real ee_obj[7];            // cd_kin_pose_invert(Twe)
};

// the scalars of DevModel the phase functions branch on, and the LDS carve-up: carried in the
// kernarg block so that a phase function has them after one scalar load (LdsLayout is declared below)
struct ModelScalars
{
   int nj, floating, tree, Sa, S, Sa_real, placed, GS, base_sph_begin, base_sph_end, jt_scan, n_static;
   int fk_split, fk_nanc, fk_b_begin, pr_rounds;
   unsigned long long live_mask, static_mask;
   unsigned long long pr_deg[2];  // a byte per round of the pair list: gather entries in use on the adding side (low nibble) and on the subtracting side (high nibble)
   int pr_hot, pad2_;
};
struct LdsLayout
{
   int T, G, W, AG, pos, ax, srad, sinact, jl, pcr, r2, end_reals;
   int Tu;                 // ORC_LDS_T_STAGED: where the update and cost-summing phases keep their copy of the trajectory (inside the tile buffers, dead by then)
   int lim_bytes;          // byte offset of the joint-limit scratch (ORC_LIM_SCRATCH bytes)
   int pstr, astr;         // waypoint strides of pos / ax: odd, so that lane = waypoint accesses (FK) hit distinct LDS banks
   int ints_bytes;         // byte offset of the int tables (slink, jtype, jcol)
   int joints_bytes;       // byte offset of the staged joint control words [nj][2]
   int sdfs_bytes;         // byte offset of the staged DevSdf[n_sdfs]
   int saff_bytes;         // byte offset of the staged affects masks [Sa]
   int sallow_bytes;       // byte offset of the self-collision partner masks [64] (robots with more than 16 active spheres)
   int ptab_bytes;         // byte offset of the staged pair list (cost_pairs.h): entries of { rsum, first | second << 8, pad } (16 bytes in fp64, 8 in fp32), then the gather words [entries][2]
   int total_bytes;
};

template <typename real>
struct DevBatch
{
   ModelScalars ms;        // = the scalars of *model
   LdsLayout lay;          // = lds_layout(...) of this launch
   const DevModel<real> * model;
   const DevSdf<real> * sdfs;
   const DevSdfCell<real> * sdfc;      // [n_sdfs] the same fields in cell units (many-sphere cost path)
   int n_sdfs;
   int n_runs, n_points, m, n;
   int tile_m;             // moving waypoints per tile (the largest tile: what the LDS carve-up holds)
   int n_tiles;            // tiles of an iteration: the first holds tile_first moving waypoints, the others tile_rest (the last what is left)
   int tile_first, tile_rest;
   // per-run state in HBM, run-major
   real * traj;            // [n_runs][n_points][n]
   real * AG;              // [n_runs][m][n]   (A^-1 G, doubles as momentum)
   real * Gdbg;            // [n_runs][m][n] or null: last gradient, for tests
   real * Gcost;           // [n_runs][m][n]: where the cost phase puts its gradient rows when !g_in_lds
   double * costs;         // [n_runs][3] total, obs, smooth
   double * trace;         // [n_runs][n_iter][3] or null
   int * status;           // [n_runs] of this launch: 0, or -1 "outside of joint limits"
   int * iters_done;       // [n_runs] iterations this launch completed (n_iter unless the run aborted)
   int * leapfrog_first;   // [n_runs]
   // run parameters
   real dt, inv_2dt, inv_dt2, lambda, inv_m;
   real epsilon, epsilon_self, obs_factor, obs_factor_self;
   int use_momentum, use_hmc, D;      // D: the derivative; -1: derivative 1 without a start boundary (tridiagonal, not Toeplitz: none of the kernels' short forms for D == 1 apply)
   // metric: band of A, endpoint couplings of B and trC
   const real * Aband;     // [2D+1][m]
   const real * beta_s;    // [m]  B[i] = beta_s[i]*q_start + beta_g[i]*q_goal
   const real * beta_g;    // [m]
   const double * metric64; // fp32 runs with derivative >= 2: Aband, beta_s, beta_g in double ([2D+1][m], [m], [m]) for the smoothness cost, whose
                           // terms (~1/dt^4) cancel to a number of order one; null otherwise
   double kss, ksg, kgg;   // trC = 0.5*(kss|s|^2 + 2 ksg s.g + kgg|g|^2)
   // A^-1 application
   int solve_mode;         // 0 cyclic reduction (tridiagonal), 1 dense A^-1, 2 closed-form Toeplitz inverse by wave scans, 3 band inverse of a higher derivative by wave scans over its generators
   int pcr_levels;
   const real * pcr;       // [levels][2][m] multipliers, then [m] inverse diagonal
   const real * Ainv;      // dense [m][m] when solve_mode == 1
   const real * jl_lo;     // [n]
   const real * jl_hi;     // [n]
   // hmc momentum resampling of this call
   const int * hmc_iters;  // [n_runs][max_resamples], -1 padded
   const real * noise;     // [n_runs][max_resamples][m][n]
   int max_resamples;
   int n_iter;
   int final_eval;
   int carry_status;       // this launch continues an iterate call (one launch per iteration: max_time, trajs_fileformstr): a run that left its joint limits in an earlier launch of the call stays out, iters_done accumulates
   long long * phase_cycles; // [n_runs][8] or null: diagnostics (cycles per phase, wave 0)
   real a_diag, a_off;     // D == 1: A = tridiag(a_off, a_diag, a_off), B couples the end rows with a_off
   int pcr_in_lds;         // the cyclic-reduction tables are staged in LDS
   int t_in_lds;           // the trajectory lives in LDS for the launch (else it is iterated in place in global memory)
   int t_staged;           // !t_in_lds: the update phase and the cost sums work on a copy in the dead tile buffers (LdsLayout::Tu)
   int g_in_lds;           // the cost phase writes its gradient rows to LDS (else to Gcost; the update phase stages them)
   int lds_flags;          // ORC_LDS_* flags of the layout
   int ag_in_lds;          // the momentum AG lives in LDS for the launch (else it is updated in place in global memory)
   int pcr_sym;            // compact tables: pcr[l][m] (towards i-s; towards i+s is the mirrored entry), then [m] inverse diagonal
   int pcr_rows;           // rows of m entries in the table
   int lim_generic;        // diagnostics: joint-limit rounds by the general (workgroup, any metric) loop
   int stagger_mode;       // 0 none; 1 odd workgroups, 2 every other group of 256: start half an iteration late
   int stagger_sleeps;     // length of that delay in s_sleep(127) units (~8k cycles each)
   // TSR hard constraints (tsr.h): n_tsrs == 0 when there are none
   const DevTsr<real> * tsrs;
   int n_tsrs, cons_k;        // constraints; rows of the system over all moving points
   int tsr_blocks;            // (constraint, point) blocks of the system
   int tsr_structured;        // the system is solved point by point (block tridiagonal KKT form, tsr.h) instead of by the dense LU
   int tsr_wcap;              // reals of the augmented block [N][N + n + 1] of that solve at its largest
   int tsr_nmax;              // N = n + (most constrained rows on one point): rows of that block at its largest
   // `start_tsr` (src/orcdchomp_mod.cpp:2316-2323, 2570-2576): the start point is a variable.  The workgroup's
   // copy of the trajectory keeps its layout [fixed row][m moving rows][goal] with an unused row in front
   // (n_points = m + 2 rows); the run's rows in global memory are np_global = m + 1: moving rows, goal.
   int free_start, np_global;
   real * tsr_ws;             // [n_runs][tsr_ws_stride] workspace: h, h0, J, J^T x, the cons_k x cons_k system
   size_t tsr_ws_stride;
   int * tsr_err;             // [n_runs] 1 after a singular system ("constraint inversion error!")
   // solve_mode 3 (derivative 2..4): the band inverse through its generators, Ainv[i][j] = sum_k U[k][i] V[k][j] for i <= j
   // (host_math.cpp build_semisep).  The tables travel where the cyclic-reduction tables of derivative 1 do (`pcr`, in LDS when
   // the plan has room), as doubles for either precision: U [ss_rank][m], V [ss_rank][m], then the band's D rows at either end,
   // [2D][2D+3] = A[i][i-D..i+D], beta_s[i], beta_g[i]
   // |D| >= 2: rows D .. m-D-1 of A are one Toeplitz row (band_c[|k|] = A[i][i+k], the same in every such row, and B is zero
   // there): the kernels take the 2D+1 coefficients from here instead of a table (host-checked: band_toeplitz)
   int band_toeplitz;
   real band_c[ORC_SS_MAX_RANK + 1];
   double band_c64[ORC_SS_MAX_RANK + 1];
   int ss_rank;
};

// Collision verdict of the trajectories of a batch (the step after the path: gettraj's re-check,
// src/orcdchomp_mod.cpp:2958-3006, with the optimizer's own sphere / field model).
template <typename real>
struct DevVerdict
{
   const DevModel<real> * model;
   const DevSdf<real> * sdfs;
   int n_sdfs;
   int n_runs, n_points, n;
   int chunk;                  // samples walked at a time (<= 64: as many as the CU's LDS holds of this robot)
   const real * traj;          // [n_runs][n_points][n]
   const int * offs;           // [n_runs+1] first sample of every run
   const int * seg;            // [samples] segment of the trajectory the sample lies on
   const real * u;             // [samples] position on the segment, 0..1
   const int * slot_xml;       // [Sa lanes] XML index of the sphere in a slot, -1: empty
   // self collision (src/orcdchomp_mod.cpp:2998-2999: `|| CheckSelfCollision`): the pairs of spheres on links that may
   // collide, XML order (a < b); an end of a pair is a slot of the position row, or -1 - k: inactive sphere k of inact_pos
   int n_pairs;
   const int * pairs;          // [n_pairs][4]: end a, end b, XML index of a, XML index of b
   const real * pair_rsum;     // [n_pairs] r_a + r_b
   const real * inact_pos;     // [inactive spheres][3] world positions
   // first contact of a run, or INT_MAX: (sample << 16) | (self << 15) | (XML sphere (a) << 8) | (field, or XML sphere b):
   // within a sample the fields come first (sphere, field order), then the pairs
   unsigned long long * key_out;   // [n_runs]: ORC_VERDICT_NONE, or sample << 32 | pair bit << 31 | XML sphere << 16 | field or partner sphere
   double * depth_out;         // [n_runs] penetration depth of that contact
};

// LDS carve-up of one workgroup (struct LdsLayout above), computed on the host (lds_layout below)
// and handed to the kernels in the kernarg block.
// Offsets are in units of `real` after a header of ORC_LDS_HEADER bytes (reduction scratch).

// what the kernels read of the robot, staged in LDS at kernel start (global reads of the
// model inside the iteration loop cost a full memory round trip each)
template <typename real>
struct ModelView
{
   int nj, n, floating, tree, Sa, S, GS, base_sph_begin, base_sph_end, jt_scan, Sa_real, placed, n_static;
   unsigned int empty_mask;                // placed row: slots without a sphere (FK keeps them at zero)
   unsigned long long live_mask;           // (the cost phase adds the lanes of static spheres: DevModel::static_mask)
   const int * slot_of;                    // [Sa_real]
   const real * base_R;                    // [9]
   const real * base_t;                    // [3]
   const int * jctl;                       // [nj][2] control words of a joint: DevJoint::packed, and aff_begin | aff_end << 8 | type << 16 | col << 24
   const real (* sph_pos)[3];              // [Sa][3]
   const unsigned long long * sph_affects; // [Sa]
   const unsigned long long * sph_allowed; // [64] bit o of entry s: sphere o is active and rides on another link than sphere s (many-sphere path)
   const __attribute__((address_space(4))) int * jpk;    // [nj] DevModel::jpacked (global memory, scalar loads)
   const __attribute__((address_space(4))) int * jpk2;   // [nj] DevModel::jpacked2
   const __attribute__((address_space(4))) real (* sph_pos_c)[3];   // DevModel::sph_pos (scalar loads: the FK walk's sphere tables)
   const __attribute__((address_space(4))) int * slot_c;            // DevModel::slot_of
   const __attribute__((address_space(4))) DevJoint<real> * joints_c;   // DevModel::joints (scalar loads: the walk's fixed transforms and axes)
   const __attribute__((address_space(4))) DevFkJoint<real> * fkj;     // DevModel::fkj (scalar loads: one record per joint)
   const __attribute__((address_space(4))) int * static_slot_c;         // DevModel::static_slot / static_pos (FK writes them into every row)
   const __attribute__((address_space(4))) real (* static_pos_c)[3];
};
#if defined(__HIPCC__)
__host__ __device__
#endif
// flags: ORC_LDS_SMALL_WORK the solve works in place (closed-form scan solve): the work buffer only
// holds the sparse joint-limit lists; ORC_LDS_G_GLOBAL the gradient rows of the cost phase go to
// global memory and the update phase keeps G in the (then dead) tile buffers
#define ORC_LDS_SMALL_WORK 1
#define ORC_LDS_G_GLOBAL   2
#define ORC_LDS_T_GLOBAL   4      // the trajectory stays in global memory between the phases (FK reads it there): the LDS then holds larger tiles
#define ORC_LDS_T_STAGED   8      // (with T_GLOBAL) the update phase and the cost sums work on a copy staged in the then dead tile buffers and write it back
inline LdsLayout lds_layout(int np, int n, int Sa, int S, int nj, int tile_m, int pcr_rows, int real_size,
   int use_ag, int n_sdfs, int sdf_size, int flags, int pair_entries = 0)
{
   const int m = np - 2, mn = m*n;
   LdsLayout L;
   int o = 0;
   auto take = [&o](int count) { const int at = o; o += (count + 3) & ~3; return at; };
   L.pstr = (Sa*3) | 1;
   L.astr = (nj*6) | 1;
   // the work buffer of the update phase (solve ping-pong, joint-limit scratch) lives in the tile
   // buffers pos/ax, which are dead by then, when they are large enough
   const int lim_reals = ((ORC_LIM_SCRATCH + real_size - 1) / real_size + 3) & ~3;
   const int work_min = 1280/real_size;
   const int work_reals = ((((flags & ORC_LDS_SMALL_WORK) || mn <= work_min) ? work_min : mn) + 3) & ~3;
   const int tile_reals = (((tile_m+2)*L.pstr + 3) & ~3) + (((tile_m+2)*L.astr + 3) & ~3);
   const bool g_global = (flags & ORC_LDS_G_GLOBAL) != 0;
   const bool t_staged = (flags & ORC_LDS_T_GLOBAL) && (flags & ORC_LDS_T_STAGED);
   const bool alias = tile_reals >= work_reals + lim_reals + (g_global ? ((mn + 3) & ~3) : 0) + (t_staged ? ((np*n + 3) & ~3) : 0);
   L.T = (flags & ORC_LDS_T_GLOBAL) ? 0 : take(np*n);
   L.G = g_global ? 0 : take(mn);
   L.W = alias ? 0 : take(work_reals);
   L.AG = take(use_ag ? mn : 0);
   L.pos = take((tile_m+2)*L.pstr);
   L.ax = take((tile_m+2)*L.astr);
   if (alias) L.W = L.pos;
   if (g_global) L.G = L.pos + work_reals + lim_reals;      // (only valid when alias: checked below)
   L.Tu = t_staged ? L.pos + work_reals + lim_reals + (g_global ? ((mn + 3) & ~3) : 0) : -1;
   L.srad = take(S);
   L.sinact = take((S-Sa)*3 + 1);
   L.jl = take(2*n);
   L.r2 = take(8*16);                      // squared ranges of the self-collision row rotations [8][16]
   L.pcr = take(pcr_rows*m);
   (void) take(Sa*3 + 12);                 // staged sphere local positions + base frame (after pcr)
   L.end_reals = o;
   L.ints_bytes = ORC_LDS_HEADER + o*real_size;
   int bytes = L.ints_bytes + (S + 2*nj + 4 + Sa) * (int) sizeof(int);     // slink, jtype, jcol, slot_of
   bytes = (bytes + 15) & ~15;
   L.joints_bytes = bytes; bytes += nj * 8; bytes = (bytes + 15) & ~15;
   L.sdfs_bytes = bytes;   bytes += n_sdfs * sdf_size; bytes = (bytes + 15) & ~15;
   L.saff_bytes = bytes;   bytes += Sa * 8;
   L.sallow_bytes = bytes; bytes += (Sa > 16) ? 64 * 8 : 0;
   bytes = (bytes + 15) & ~15;
   L.ptab_bytes = bytes;   bytes += pair_entries * (2 * real_size + 8);
   if (alias) L.lim_bytes = ORC_LDS_HEADER + (L.pos + work_reals) * real_size;
   else { L.lim_bytes = bytes; bytes += ORC_LIM_SCRATCH; }
   L.total_bytes = ((g_global || t_staged) && !alias) ? (1 << 30) : bytes;      // G (and the staged trajectory) in the tile buffers need tiles that hold them
   return L;
}
