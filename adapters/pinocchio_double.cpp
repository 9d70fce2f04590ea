// pinocchio_double.cpp -- fills the reference's empty slot test/pinocchio_double.cpp (0 bytes): the explicit instantiation
// `template struct ddp::pinocchio::model_t<double>;` (seam B2), backed by this repository's own rigid-body
// code through the C-ABI instead of Pinocchio.  Members are declared at include/ddp/pinocchio_model.hpp:77-186; the
// reference defines them in pinocchio_model.ipp (GENERAL :42-326, ABA :328-405, FRAMES :407-481) on top of Pinocchio.
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (needs the reference's headers: Eigen, Boost, fmt).  Written against the
// declarations as they stand; tests/test_adapters.py checks every ddp_hip_* call against include/ddp_hip/ddp_hip.h.
// Scope: vector-space configurations (trees of 1-DoF joints, nq == nv -- what the reference's FD mode 1 requires,
// problem.hpp:78-81).  The host-side members evaluate ONE configuration per call on the device (ddp_hip_model_*): they keep
// the reference's CPU code paths (make_trajectory, cost_seq_aug, the self checks) working; the hot path does not go
// through them but through the batched entry points (adapters/ddp/ddp_bwd.ipp, ddp_fwd.ipp, hip_bridge::compute_derivatives_resident).
#include "ddp/pinocchio_model.hpp"

#include "ddp_hip/ddp_hip.h"
#include "urdf_reader.hpp"

#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>

namespace ddp {
namespace hip_bridge {
void model_table_of(void const* model, ddp_hip_model& out, ddp_hip_model_storage& st);
void frame_of(void const* model, index_t frame_id, int32_t& joint, double off[3]);
}  // namespace hip_bridge
namespace pinocchio {

template <>
struct model_t<double>::impl_model_t {
  ddp_hip_urdf::model_table table;
  ddp_hip_model_handle* handle = nullptr;
  std::string name;
};
template <>
struct model_t<double>::impl_data_t {};   // the library keeps its own scratch; nothing per thread on the host

namespace {
void ok(int rc, char const* where) {
  if (rc < 0) throw std::runtime_error(std::string(where) + ": " + ddp_hip_strerror(rc));
}
// model_t keeps its implementation private and its header (which the drop-in must not touch) declares no accessor: the
// tables are found again by the object's address.  model_t is move-only; the move constructor re-registers.
std::map<void const*, ddp_hip_urdf::model_table const*>& registry() {
  static std::map<void const*, ddp_hip_urdf::model_table const*> r;
  return r;
}
std::mutex& registry_mutex() {
  static std::mutex m;
  return m;
}
}  // namespace
}  // namespace pinocchio

namespace hip_bridge {
// declared in adapters/ddp_hip_bridge.hpp: the table ddp_hip_create needs, and the joint / offset a frame index means
void model_table_of(void const* model, ddp_hip_model& out, ddp_hip_model_storage& st) {
  std::lock_guard<std::mutex> lock(pinocchio::registry_mutex());
  auto it = pinocchio::registry().find(model);
  if (it == pinocchio::registry().end()) throw std::invalid_argument("not a model_t<double> of adapters/pinocchio_double.cpp");
  st = it->second->st;
  out = it->second->model;
  out.parent = st.parent; out.jtype = st.jtype; out.axis = st.axis; out.Rp = st.Rp; out.pp = st.pp;
  out.mass_j = st.mass_j; out.com = st.com; out.Ic = st.Ic;
}
void frame_of(void const* model, index_t frame_id, int32_t& joint, double off[3]) {
  std::lock_guard<std::mutex> lock(pinocchio::registry_mutex());
  auto it = pinocchio::registry().find(model);
  if (it == pinocchio::registry().end()) throw std::invalid_argument("not a model_t<double> of adapters/pinocchio_double.cpp");
  index_t k = 0;                 // the reference addresses frames by Pinocchio's frame index; here: the links of the URDF in name order
  for (auto const& kv : it->second->frames) {
    if (k++ == frame_id) { joint = kv.second.first; for (int c = 0; c < 3; ++c) off[c] = kv.second.second.v[c]; return; }
  }
  throw std::out_of_range("frame index");
}
}  // namespace hip_bridge

namespace pinocchio {

// ctor from a URDF path (pinocchio_model.ipp:155-156); `~` expands to $HOME like :101-108
template <>
model_t<double>::model_t(fmt::string_view urdf_path, index_t n_parallel) noexcept(false) {
  if (n_parallel != 1) throw std::invalid_argument("n_parallel must be 1 (pinocchio_model.ipp:121)");
  std::string path{urdf_path.begin(), urdf_path.end()};
  if (!path.empty() && path[0] == '~') path = std::string(std::getenv("HOME")) + path.substr(1);
  m_model = new impl_model_t{};
  m_data = new impl_data_t{};
  m_num_data = 1;
  ddp_hip_urdf::build_from_file(path, /*floating_base=*/false, m_model->table);
  m_model->name = path;
  ok(ddp_hip_model_create(&m_model->table.model, /*device=*/0, &m_model->handle), "ddp_hip_model_create");
  m_config_dim = m_model->table.model.nv;
  m_tangent_dim = m_model->table.model.nv;
  std::lock_guard<std::mutex> lock(registry_mutex());
  registry()[this] = &m_model->table;
}

template <>
model_t<double>::~model_t() noexcept {
  { std::lock_guard<std::mutex> lock(registry_mutex()); registry().erase(this); }
  if (m_model) { if (m_model->handle) ddp_hip_model_destroy(m_model->handle); delete m_model; }
  delete m_data;
}
template <>
model_t<double>::model_t(model_t&& other) noexcept
    : m_model(other.m_model), m_data(other.m_data), m_num_data(other.m_num_data), m_config_dim(other.m_config_dim),
      m_tangent_dim(other.m_tangent_dim) {
  other.m_model = nullptr; other.m_data = nullptr; other.m_num_data = 0;
  std::lock_guard<std::mutex> lock(registry_mutex());
  registry().erase(&other);
  if (m_model) registry()[this] = &m_model->table;
}
template <>
auto model_t<double>::get_data() const noexcept -> impl_data_t* { return m_data; }
template <>
auto model_t<double>::model_name() const noexcept -> fmt::string_view { return m_model->name; }

// ---- GENERAL (pinocchio_model.ipp:42-326): vector-space configurations ---------------------------------------------------
template <>
void model_t<double>::neutral_configuration(mut_view_t<Eigen::Dynamic> out_q) const noexcept { out_q.setZero(); }
template <>
void model_t<double>::random_configuration(mut_view_t<Eigen::Dynamic> out_q) const noexcept { out_q.setRandom(); }
template <>
void model_t<double>::integrate(mut_view_t<Eigen::Dynamic> out_q, const_view_t<Eigen::Dynamic> q, const_view_t<Eigen::Dynamic> v) const noexcept {
  out_q = q + v;                                                                  // :222-236 on a vector space
}
template <>
void model_t<double>::d_integrate_dq(mut_view_t<Eigen::Dynamic, Eigen::Dynamic> out, const_view_t<Eigen::Dynamic>, const_view_t<Eigen::Dynamic>) const noexcept {
  out.setIdentity();                                                              // :238-252
}
template <>
void model_t<double>::d_integrate_dv(mut_view_t<Eigen::Dynamic, Eigen::Dynamic> out, const_view_t<Eigen::Dynamic>, const_view_t<Eigen::Dynamic>) const noexcept {
  out.setIdentity();                                                              // :254-269
}
template <>
void model_t<double>::difference(mut_view_t<Eigen::Dynamic> out_v, const_view_t<Eigen::Dynamic> q_start, const_view_t<Eigen::Dynamic> q_finish) const noexcept {
  out_v = q_finish - q_start;                                                     // :271-286
}
template <>
void model_t<double>::d_difference_dq_start(mut_view_t<Eigen::Dynamic, Eigen::Dynamic> out, const_view_t<Eigen::Dynamic>, const_view_t<Eigen::Dynamic>) const noexcept {
  out.setIdentity(); out *= -1.0;                                                 // :288-304
}
template <>
void model_t<double>::d_difference_dq_finish(mut_view_t<Eigen::Dynamic, Eigen::Dynamic> out, const_view_t<Eigen::Dynamic>, const_view_t<Eigen::Dynamic>) const noexcept {
  out.setIdentity();                                                              // :306-321
}

// ---- ABA (pinocchio_model.ipp:328-405) --------------------------------------------------------------------------------------
template <>
void model_t<double>::dynamics_aba(mut_view_t<Eigen::Dynamic> out_acceleration, const_view_t<Eigen::Dynamic> q, const_view_t<Eigen::Dynamic> v,
                                   const_view_t<Eigen::Dynamic> tau) const noexcept {
  int rc = ddp_hip_model_aba(m_model->handle, q.data(), v.data(), tau.data(), out_acceleration.data());
  DDP_ASSERT_MSG(ddp_hip_strerror(rc), rc >= 0);
}
template <>
void model_t<double>::d_dynamics_aba(mut_view_t<Eigen::Dynamic, Eigen::Dynamic> out_dq, mut_view_t<Eigen::Dynamic, Eigen::Dynamic> out_dv,
                                     mut_view_t<Eigen::Dynamic, Eigen::Dynamic> out_dtau, const_view_t<Eigen::Dynamic> q,
                                     const_view_t<Eigen::Dynamic> v, const_view_t<Eigen::Dynamic> tau) const noexcept {
  // the views are column-major with an outer stride (utils.hpp:265-266): marshal through contiguous nv x nv buffers
  const index_t nv = m_tangent_dim;
  Eigen::MatrixXd dq(nv, nv), dv(nv, nv), dt(nv, nv);
  int rc = ddp_hip_model_aba_derivatives(m_model->handle, q.data(), v.data(), tau.data(), dq.data(), dv.data(), dt.data());
  DDP_ASSERT_MSG(ddp_hip_strerror(rc), rc >= 0);
  out_dq = dq; out_dv = dv; out_dtau = dt;
}

// ---- FRAMES (pinocchio_model.ipp:407-481) -----------------------------------------------------------------------------------
template <>
auto model_t<double>::frame_coordinates(index_t i, const_view_t<Eigen::Dynamic> q) const noexcept -> Eigen::Matrix<double, 3, 1> {
  int32_t joint; double off[3], p[3];
  hip_bridge::frame_of(this, i, joint, off);
  int rc = ddp_hip_model_frame(m_model->handle, joint, off, q.data(), p, nullptr);
  DDP_ASSERT_MSG(ddp_hip_strerror(rc), rc >= 0);
  return {p[0], p[1], p[2]};
}
template <>
void model_t<double>::d_frame_coordinates(mut_view_t<3, Eigen::Dynamic> out, index_t i, const_view_t<Eigen::Dynamic> q) const noexcept {
  int32_t joint; double off[3], p[3];
  hip_bridge::frame_of(this, i, joint, off);
  Eigen::Matrix<double, 3, Eigen::Dynamic> J(3, m_tangent_dim);
  int rc = ddp_hip_model_frame(m_model->handle, joint, off, q.data(), p, J.data());   // the reference's WORLD-frame rows (:458-461)
  DDP_ASSERT_MSG(ddp_hip_strerror(rc), rc >= 0);
  out = J;
}
template <>
auto model_t<double>::n_frames() const noexcept -> index_t { return static_cast<index_t>(m_model->table.frames.size()); }
template <>
auto model_t<double>::frame_name(index_t i) const noexcept -> fmt::string_view {
  index_t k = 0;
  for (auto const& kv : m_model->table.frames) if (k++ == i) return kv.first;
  return "";
}

template struct model_t<double>;   // test/pinocchio_mpfr_500.cpp:9-16 does the same for the mpfr scalar

}  // namespace pinocchio
}  // namespace ddp
