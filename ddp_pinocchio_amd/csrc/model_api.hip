// model_api.hip -- point evaluations of the Model concept (pinocchio_model.hpp:77-186) on the device, for the B2 seam:
// a host-side ddp::pinocchio::model_t<double> (adapters/pinocchio_double.cpp) answers dynamics_aba / d_dynamics_aba /
// frame_coordinates / d_frame_coordinates for ONE configuration at a time through these calls.  They are plumbing (one
// lane, one launch and two copies per call); the hot path uses the batched entry points (ddp_hip_linearize, ...).
#include <math.h>
#include <string.h>

#include <new>

#include "internal.h"
#include "rbd.h"
#include "rbd_deriv.h"

extern void ddp_hip_fill_dev_model(const ddp_hip_model* mo, DevModel& dm);   // ctx.hip
extern bool ddp_hip_build_tables(DevModel& dm);

struct ddp_hip_model_handle {
  int device = 0;
  DevModel model_h{};
  DevModel* model_d = nullptr;
  double* buf_d = nullptr;     // [in nq + 2 nv | out 3 nv nv + 4 nv]
  hipStream_t stream = nullptr;
};

namespace {

template <int NJ>
__global__ void model_aba_kernel(const DevModel* m, const double* in, double* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int nv = m->nv;
  if (m->kind == DDP_HIP_MODEL_PENDULUM) { out[0] = rbd::pendulum_acc(*m, in[0], in[2]); return; }
  const int nq = m->nq;
  double q[NJ + 1], v[NJ], tau[NJ], a[NJ];
  for (int i = 0; i < nq; ++i) q[i] = in[i];
  for (int i = 0; i < nv; ++i) { v[i] = in[nq + i]; tau[i] = in[nq + nv + i]; }
  rbd::aba_tree<NJ>(*m, q, v, tau, a);
  for (int i = 0; i < nv; ++i) out[i] = a[i];
}

template <int NJ>
__global__ void model_aba_deriv_kernel(const DevModel* m, const double* in, double* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int nv = m->nv;
  if (m->kind == DDP_HIP_MODEL_PENDULUM) {      // pendulum_model.hpp:116-130
    out[0] = -9.81 / m->length * cos(in[0]); out[1] = 0.0; out[2] = 1.0 / m->mass;
    return;
  }
  double q[NJ], v[NJ], tau[NJ], a[NJ];
  for (int i = 0; i < nv; ++i) { q[i] = in[i]; v[i] = in[nv + i]; tau[i] = in[2 * nv + i]; }
  rbdd::aba_derivatives_lane<NJ>(*m, q, v, tau, a, out, out + nv * nv, out + 2 * nv * nv);
}

template <int NJ>
__global__ void model_frame_kernel(const DevModel* m, const double* in, double* out, int want_jac) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double q[NJ + 1];
  for (int i = 0; i < m->nq; ++i) q[i] = in[i];
  rbd::frame_position<NJ>(*m, q, out, want_jac ? out + 3 : nullptr);
}

#define MODEL_DISPATCH(nv, CALL)       \
  do {                                 \
    if ((nv) <= 6) { CALL(6); }        \
    else if ((nv) <= 38) { CALL(38); } \
    else { CALL(64); }                 \
  } while (0)

int run(ddp_hip_model_handle* h, const double* in, size_t n_in, double* out, size_t n_out, int what, int want_jac) {
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpyAsync(h->buf_d, in, sizeof(double) * n_in, hipMemcpyHostToDevice, h->stream));
  double* o = h->buf_d + 3 * DDP_MAXJ + 8;
  const int nv = h->model_h.nv;
  if (what == 0) {
#define CALL(NJ) hipLaunchKernelGGL((model_aba_kernel<NJ>), dim3(1), dim3(64), 0, h->stream, h->model_d, h->buf_d, o)
    MODEL_DISPATCH(nv, CALL);
#undef CALL
  } else if (what == 1) {
    // one lane holds the whole recursion in private memory: 38 joints is what its frame allows (the batched kernels of
    // lin_analytic.hip have no such limit)
    if (nv <= 6) hipLaunchKernelGGL((model_aba_deriv_kernel<6>), dim3(1), dim3(64), 0, h->stream, h->model_d, h->buf_d, o);
    else if (nv <= 38) hipLaunchKernelGGL((model_aba_deriv_kernel<38>), dim3(1), dim3(64), 0, h->stream, h->model_d, h->buf_d, o);
    else return DDP_HIP_E_UNSUPPORTED;
  } else {
#define CALL(NJ) hipLaunchKernelGGL((model_frame_kernel<NJ>), dim3(1), dim3(64), 0, h->stream, h->model_d, h->buf_d, o, want_jac)
    MODEL_DISPATCH(nv, CALL);
#undef CALL
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, o, sizeof(double) * n_out, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return DDP_HIP_OK;
}

}  // namespace

extern "C" int ddp_hip_model_create(const ddp_hip_model* model, int device, ddp_hip_model_handle** out) {
  if (!model || !out) return DDP_HIP_E_ARG;
  *out = nullptr;
  if (model->nv < 1 || model->nv > DDP_MAXJ) return DDP_HIP_E_ARG;
  if (model->kind == DDP_HIP_MODEL_TREE) {
    if (!model->parent || !model->jtype || !model->axis || !model->Rp || !model->pp || !model->mass_j || !model->com || !model->Ic) return DDP_HIP_E_ARG;
    const int nj = model->jtype[0] == DDP_HIP_JOINT_FREEFLYER ? model->nv - 5 : model->nv;
    if (nj < 1) return DDP_HIP_E_ARG;
    for (int i = 0; i < nj; ++i)
      if (model->parent[i] >= i || model->parent[i] < -1) return DDP_HIP_E_ARG;
  } else if (model->kind != DDP_HIP_MODEL_PENDULUM) return DDP_HIP_E_ARG;
  int ndev = ddp_hip_device_count();
  if (ndev <= 0) return DDP_HIP_E_NODEVICE;
  if (device < 0 || device >= ndev) return DDP_HIP_E_ARG;
  HIP_TRY(hipSetDevice(device));
  ddp_hip_model_handle* h = new (std::nothrow) ddp_hip_model_handle();
  if (!h) return DDP_HIP_E_HIP;
  h->device = device;
  ddp_hip_fill_dev_model(model, h->model_h);
  if (model->kind == DDP_HIP_MODEL_TREE && !ddp_hip_build_tables(h->model_h)) { delete h; return DDP_HIP_E_UNSUPPORTED; }
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(&h->model_d, sizeof(DevModel)) != hipSuccess ||
      hipMalloc(&h->buf_d, sizeof(double) * (size_t)(3 * DDP_MAXJ + 8 + 3 * DDP_MAXJ * DDP_MAXJ + 4 * DDP_MAXJ)) != hipSuccess ||
      hipMemcpy(h->model_d, &h->model_h, sizeof(DevModel), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    ddp_hip_model_destroy(h);
    return DDP_HIP_E_HIP;
  }
  *out = h;
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_model_destroy(ddp_hip_model_handle* h) {
  if (!h) return DDP_HIP_E_ARG;
  (void)hipSetDevice(h->device);
  if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
  if (h->model_d) (void)hipFree(h->model_d);
  if (h->buf_d) (void)hipFree(h->buf_d);
  delete h;
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_model_aba(ddp_hip_model_handle* h, const double* q, const double* v, const double* tau, double* qdd) {
  if (!h || !q || !v || !tau || !qdd) return DDP_HIP_E_ARG;
  const int nv = h->model_h.nv, nq = h->model_h.nq;
  double in[3 * DDP_MAXJ + 1];
  memcpy(in, q, sizeof(double) * nq); memcpy(in + nq, v, sizeof(double) * nv); memcpy(in + nq + nv, tau, sizeof(double) * nv);
  return run(h, in, (size_t)(nq + 2 * nv), qdd, (size_t)nv, 0, 0);
}

extern "C" int ddp_hip_model_aba_derivatives(ddp_hip_model_handle* h, const double* q, const double* v, const double* tau,
                                             double* dq, double* dv, double* dtau) {
  if (!h || !q || !v || !tau || !dq || !dv || !dtau) return DDP_HIP_E_ARG;
  if (h->model_h.ff) return DDP_HIP_E_UNSUPPORTED;        // analytic partials: vector-space configurations (as the reference's mode 1)
  const int nv = h->model_h.nv;
  double in[3 * DDP_MAXJ];
  memcpy(in, q, sizeof(double) * nv); memcpy(in + nv, v, sizeof(double) * nv); memcpy(in + 2 * nv, tau, sizeof(double) * nv);
  const size_t nn = (size_t)nv * nv;
  double* out = new (std::nothrow) double[3 * nn];
  if (!out) return DDP_HIP_E_HIP;
  const int rc = run(h, in, 3 * (size_t)nv, out, 3 * nn, 1, 0);
  if (rc == DDP_HIP_OK) { memcpy(dq, out, sizeof(double) * nn); memcpy(dv, out + nn, sizeof(double) * nn); memcpy(dtau, out + 2 * nn, sizeof(double) * nn); }
  delete[] out;
  return rc;
}

extern "C" int ddp_hip_model_frame(ddp_hip_model_handle* h, int32_t joint, const double off[3], const double* q, double* p3, double* J) {
  if (!h || !off || !q || !p3 || h->model_h.kind != DDP_HIP_MODEL_TREE || joint < 0 || joint >= h->model_h.nj) return DDP_HIP_E_ARG;
  const int nv = h->model_h.nv;
  HIP_TRY(hipSetDevice(h->device));
  // the frame is part of the (small) device model: patch it for this call
  h->model_h.frame_joint = joint;
  for (int k = 0; k < 3; ++k) h->model_h.frame_off[k] = off[k];
  HIP_TRY(hipMemcpyAsync(h->model_d, &h->model_h, sizeof(DevModel), hipMemcpyHostToDevice, h->stream));
  double out[3 + 3 * DDP_MAXJ];
  const int rc = run(h, q, (size_t)h->model_h.nq, out, (size_t)(3 + (J ? 3 * nv : 0)), 2, J ? 1 : 0);
  if (rc != DDP_HIP_OK) return rc;
  memcpy(p3, out, sizeof(double) * 3);
  if (J) memcpy(J, out + 3, sizeof(double) * 3 * nv);
  return DDP_HIP_OK;
}
