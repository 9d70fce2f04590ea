// models.cpp -- built-in, seeded model tables.  The reference loads URDF files through Pinocchio
// (pinocchio_model.ipp:98-113); neither urdfdom nor any .urdf exists offline, so the
// robots of BASELINE.json are hard-coded here: the 1-DoF pendulum of test/pendulum_ddp.cpp:30, a
// 6-revolute chain with UR5-like dimensions, and a 38-DoF tree with the topology of Talos
// (3 prismatic + 3 revolute base joints, 2 legs x 6, torso 2, 2 arms x 8, head 2) whose inertial
// parameters are drawn from the seed.
#include <math.h>
#include <string.h>

#include "ddp_hip/ddp_hip.h"

namespace {

struct SplitMix64 {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform(double lo, double hi) {
    double u = (double)(next() >> 11) * (1.0 / 9007199254740992.0);
    return lo + (hi - lo) * u;
  }
};

void mat3_mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
void rot_xyz(double rx, double ry, double rz, double* R) {
  double cx = cos(rx), sx = sin(rx), cy = cos(ry), sy = sin(ry), cz = cos(rz), sz = sin(rz);
  double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
  double Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
  double Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  double t[9];
  mat3_mul(Ry, Rx, t);
  mat3_mul(Rz, t, R);
}
void set_identity(double* R) {
  memset(R, 0, 9 * sizeof(double));
  R[0] = R[4] = R[8] = 1.0;
}
// rotational inertia of a solid box (a,b,c) of mass m, rotated by R: R diag R^T
void box_inertia(double m, double a, double b, double c, const double* R, double* I) {
  double d[3] = {m * (b * b + c * c) / 12.0, m * (a * a + c * c) / 12.0, m * (a * a + b * b) / 12.0};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += R[3 * i + k] * d[k] * R[3 * j + k];
      I[3 * i + j] = s;
    }
}
void set_axis(double* a, char which) {
  a[0] = a[1] = a[2] = 0;
  a[which == 'x' ? 0 : which == 'y' ? 1 : 2] = 1.0;
}

void point_model(ddp_hip_model_storage* st, ddp_hip_model* out) {
  out->parent = st->parent; out->jtype = st->jtype; out->axis = st->axis; out->Rp = st->Rp; out->pp = st->pp;
  out->mass_j = st->mass_j; out->com = st->com; out->Ic = st->Ic;
}

void build_chain6(ddp_hip_model_storage* st, ddp_hip_model* out) {
  // UR5-like: shoulder pan (z), shoulder lift (y), elbow (y), wrist 1 (y), wrist 2 (z), wrist 3 (y)
  const char axes[6] = {'z', 'y', 'y', 'y', 'z', 'y'};
  const double off[6][3] = {{0, 0, 0.089159}, {0, 0.13585, 0}, {0, -0.1197, 0.425},
                            {0, 0, 0.39225},  {0, 0.093, 0},   {0, 0, 0.09465}};
  const double mass[6] = {3.7, 8.393, 2.275, 1.219, 1.219, 0.1879};
  const double com[6][3] = {{0, 0, 0}, {0, 0, 0.28}, {0, 0, 0.25}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  const double box[6][3] = {{0.12, 0.12, 0.15}, {0.1, 0.1, 0.5}, {0.08, 0.08, 0.45},
                            {0.08, 0.08, 0.1},  {0.08, 0.08, 0.1}, {0.06, 0.06, 0.04}};
  double Id[9];
  set_identity(Id);
  for (int i = 0; i < 6; ++i) {
    st->parent[i] = i - 1;
    st->jtype[i] = DDP_HIP_JOINT_REVOLUTE;
    set_axis(st->axis + 3 * i, axes[i]);
    set_identity(st->Rp + 9 * i);
    for (int k = 0; k < 3; ++k) { st->pp[3 * i + k] = off[i][k]; st->com[3 * i + k] = com[i][k]; }
    st->mass_j[i] = mass[i];
    box_inertia(mass[i], box[i][0], box[i][1], box[i][2], Id, st->Ic + 9 * i);
  }
  out->kind = DDP_HIP_MODEL_TREE;
  out->nv = 6;
  out->mass = out->length = 0;
  out->gravity[0] = 0; out->gravity[1] = 0; out->gravity[2] = -9.81;
  point_model(st, out);
}

// free_flyer: the pelvis hangs from ONE SE(3) joint (q = [p, quaternion], nq = 39, nv = 38: the real Talos layout)
// instead of 3 prismatic + 3 revolute joints (nq = nv = 38, a vector space)
// the UR5-like chain on a floating base of 10 kg (nq = 13, nv = 12): the small Lie-group test model
void build_chain6_ff(ddp_hip_model_storage* st, ddp_hip_model* out) {
  ddp_hip_model tmp;
  ddp_hip_model_storage arm;
  memset(&arm, 0, sizeof(arm));
  build_chain6(&arm, &tmp);
  double Id[9];
  set_identity(Id);
  st->parent[0] = -1; st->jtype[0] = DDP_HIP_JOINT_FREEFLYER;
  set_axis(st->axis, 'x'); set_identity(st->Rp);
  st->mass_j[0] = 10.0; st->com[0] = 0.01; st->com[1] = -0.02; st->com[2] = 0.03;
  box_inertia(10.0, 0.4, 0.3, 0.2, Id, st->Ic);
  for (int i = 0; i < 6; ++i) {
    st->parent[i + 1] = i; st->jtype[i + 1] = arm.jtype[i];
    memcpy(st->axis + 3 * (i + 1), arm.axis + 3 * i, 3 * sizeof(double));
    memcpy(st->Rp + 9 * (i + 1), arm.Rp + 9 * i, 9 * sizeof(double));
    memcpy(st->pp + 3 * (i + 1), arm.pp + 3 * i, 3 * sizeof(double));
    st->mass_j[i + 1] = arm.mass_j[i];
    memcpy(st->com + 3 * (i + 1), arm.com + 3 * i, 3 * sizeof(double));
    memcpy(st->Ic + 9 * (i + 1), arm.Ic + 9 * i, 9 * sizeof(double));
  }
  out->kind = DDP_HIP_MODEL_TREE;
  out->nv = 12;
  out->mass = out->length = 0;
  out->gravity[0] = 0; out->gravity[1] = 0; out->gravity[2] = -9.81;
  point_model(st, out);
}

void build_tree38(uint64_t seed, ddp_hip_model_storage* st, ddp_hip_model* out, bool free_flyer = false) {
  SplitMix64 rng{seed};
  int j = 0;
  auto add = [&](int parent, int type, char axis, double px, double py, double pz, bool massless) {
    st->parent[j] = parent;
    st->jtype[j] = type;
    set_axis(st->axis + 3 * j, axis);
    double R[9];
    if (massless) set_identity(R);
    else rot_xyz(rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), R);
    memcpy(st->Rp + 9 * j, R, sizeof(R));
    st->pp[3 * j + 0] = px; st->pp[3 * j + 1] = py; st->pp[3 * j + 2] = pz;
    if (massless) {
      st->mass_j[j] = 0;
      memset(st->com + 3 * j, 0, 3 * sizeof(double));
      memset(st->Ic + 9 * j, 0, 9 * sizeof(double));
    } else {
      double m = rng.uniform(0.5, 5.0);
      st->mass_j[j] = m;
      for (int k = 0; k < 3; ++k) st->com[3 * j + k] = rng.uniform(-0.05, 0.05);
      double Rb[9];
      rot_xyz(rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), Rb);
      box_inertia(m, rng.uniform(0.05, 0.3), rng.uniform(0.05, 0.3), rng.uniform(0.05, 0.3), Rb, st->Ic + 9 * j);
    }
    return j++;
  };
  auto len = [&]() { return rng.uniform(0.05, 0.3); };
  int pelvis;
  if (free_flyer) {
    pelvis = add(-1, DDP_HIP_JOINT_FREEFLYER, 'x', 0, 0, 0, false);
    set_identity(st->Rp);                     // the placement of a free flyer IS its configuration
  } else {
    // floating base as 3 prismatic + 3 revolute joints; the first five carry no mass, the sixth is the pelvis
    int b = add(-1, DDP_HIP_JOINT_PRISMATIC, 'x', 0, 0, 0, true);
    b = add(b, DDP_HIP_JOINT_PRISMATIC, 'y', 0, 0, 0, true);
    b = add(b, DDP_HIP_JOINT_PRISMATIC, 'z', 0, 0, 1.0, true);
    b = add(b, DDP_HIP_JOINT_REVOLUTE, 'z', 0, 0, 0, true);
    b = add(b, DDP_HIP_JOINT_REVOLUTE, 'y', 0, 0, 0, true);
    pelvis = add(b, DDP_HIP_JOINT_REVOLUTE, 'x', 0, 0, 0, false);
  }
  const char leg_axes[6] = {'z', 'x', 'y', 'y', 'y', 'x'};
  for (int side = 0; side < 2; ++side) {
    double sy = side == 0 ? 1.0 : -1.0;
    int p = pelvis;
    for (int k = 0; k < 6; ++k) p = add(p, DDP_HIP_JOINT_REVOLUTE, leg_axes[k], 0, k == 0 ? sy * len() : 0, -len(), false);
  }
  int torso = add(pelvis, DDP_HIP_JOINT_REVOLUTE, 'z', 0, 0, len(), false);
  torso = add(torso, DDP_HIP_JOINT_REVOLUTE, 'y', 0, 0, len(), false);
  const char arm_axes[8] = {'y', 'x', 'z', 'y', 'z', 'y', 'x', 'z'};
  for (int side = 0; side < 2; ++side) {
    double sy = side == 0 ? 1.0 : -1.0;
    int p = torso;
    for (int k = 0; k < 8; ++k) p = add(p, DDP_HIP_JOINT_REVOLUTE, arm_axes[k], 0, sy * len(), k == 0 ? len() : 0, false);
  }
  int head = add(torso, DDP_HIP_JOINT_REVOLUTE, 'y', 0, 0, len(), false);
  add(head, DDP_HIP_JOINT_REVOLUTE, 'z', 0, 0, len(), false);
  out->kind = DDP_HIP_MODEL_TREE;
  out->nv = free_flyer ? j + 5 : j;  // 38 either way: 33 joints, six of the velocities belong to the root
  out->mass = out->length = 0;
  out->gravity[0] = 0; out->gravity[1] = 0; out->gravity[2] = -9.81;
  point_model(st, out);
}

}  // namespace

extern "C" int ddp_hip_builtin_model(int which, uint64_t seed, ddp_hip_model_storage* storage, ddp_hip_model* out) {
  if (!storage || !out) return DDP_HIP_E_ARG;
  memset(storage, 0, sizeof(*storage));
  memset(out, 0, sizeof(*out));
  switch (which) {
    case DDP_HIP_BUILTIN_PENDULUM:
      out->kind = DDP_HIP_MODEL_PENDULUM;
      out->nv = 1;
      out->mass = 1.0;    // test/pendulum_ddp.cpp:30
      out->length = 1.0;
      out->gravity[2] = -9.81;
      point_model(storage, out);
      return DDP_HIP_OK;
    case DDP_HIP_BUILTIN_CHAIN6:
      build_chain6(storage, out);
      return DDP_HIP_OK;
    case DDP_HIP_BUILTIN_TREE38:
      build_tree38(seed, storage, out);
      return DDP_HIP_OK;
    case DDP_HIP_BUILTIN_CHAIN6_FF:
      build_chain6_ff(storage, out);
      return DDP_HIP_OK;
    case DDP_HIP_BUILTIN_TREE38_FF:
      build_tree38(seed, storage, out, true);
      return DDP_HIP_OK;
    default:
      return DDP_HIP_E_ARG;
  }
}
