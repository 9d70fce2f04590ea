// urdf_reader.hpp -- minimal URDF -> ddp_hip_model table (dependency-free C++17; compiled and tested in this repository:
// tests/test_adapters.py builds host/test_urdf.cpp with g++).  Stands in for pinocchio::urdf::buildModel at
// pinocchio_model.ipp:98-113 for the robots the hot path supports: trees of 1-DoF joints (revolute / continuous /
// prismatic) + fixed joints, which are merged into their parent body like Pinocchio does.  A floating base is modelled as
// 3 prismatic + 3 revolute joints when `floating_base` is set.  Not supported (an exception says so):
// planar / floating / spherical URDF joints, mimic joints, xacro.
#ifndef DDP_HIP_URDF_READER_HPP
#define DDP_HIP_URDF_READER_HPP

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "ddp_hip/ddp_hip.h"

namespace ddp_hip_urdf {

struct vec3 { double v[3] = {0, 0, 0}; };
struct mat3 { double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; };   // row-major

inline mat3 mul(const mat3& a, const mat3& b) {
  mat3 c;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return c;
}
inline vec3 mul(const mat3& a, const vec3& x) {
  vec3 y;
  for (int i = 0; i < 3; ++i) y.v[i] = a.m[3 * i] * x.v[0] + a.m[3 * i + 1] * x.v[1] + a.m[3 * i + 2] * x.v[2];
  return y;
}
inline mat3 transpose(const mat3& a) {
  mat3 t;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t.m[3 * i + j] = a.m[3 * j + i];
  return t;
}
// URDF rpy: R = Rz(yaw) Ry(pitch) Rx(roll)
inline mat3 rpy(double r, double p, double y) {
  const double cr = std::cos(r), sr = std::sin(r), cp = std::cos(p), sp = std::sin(p), cy = std::cos(y), sy = std::sin(y);
  mat3 R;
  R.m[0] = cy * cp; R.m[1] = cy * sp * sr - sy * cr; R.m[2] = cy * sp * cr + sy * sr;
  R.m[3] = sy * cp; R.m[4] = sy * sp * sr + cy * cr; R.m[5] = sy * sp * cr - cy * sr;
  R.m[6] = -sp;     R.m[7] = cp * sr;                R.m[8] = cp * cr;
  return R;
}

// ---- a very small XML reader: elements, attributes, no entities / CDATA ------------------------------------------------
struct xml_node {
  std::string name;
  std::map<std::string, std::string> attr;
  std::vector<xml_node> children;
  const xml_node* child(const std::string& n) const {
    for (auto const& c : children) if (c.name == n) return &c;
    return nullptr;
  }
  std::string get(const std::string& k, const std::string& dflt = "") const {
    auto it = attr.find(k);
    return it == attr.end() ? dflt : it->second;
  }
};

class xml_parser {
  const std::string& s;
  size_t i = 0;
  void skip_ws() { while (i < s.size() && std::isspace(static_cast<unsigned char>(s[i]))) ++i; }
  bool starts(const char* t) const { return s.compare(i, std::strlen(t), t) == 0; }
  void skip_misc() {
    for (;;) {
      skip_ws();
      if (starts("<?")) { i = s.find("?>", i); if (i == std::string::npos) throw std::runtime_error("urdf: unterminated <?"); i += 2; }
      else if (starts("<!--")) { i = s.find("-->", i); if (i == std::string::npos) throw std::runtime_error("urdf: unterminated comment"); i += 3; }
      else if (starts("<!")) { i = s.find('>', i); if (i == std::string::npos) throw std::runtime_error("urdf: unterminated <!"); i += 1; }
      else return;
    }
  }
  std::string name() {
    size_t b = i;
    while (i < s.size() && (std::isalnum(static_cast<unsigned char>(s[i])) || s[i] == '_' || s[i] == ':' || s[i] == '-' || s[i] == '.')) ++i;
    if (i == b) throw std::runtime_error("urdf: expected a name");
    return s.substr(b, i - b);
  }
 public:
  explicit xml_parser(const std::string& text) : s(text) {}
  xml_node element() {
    skip_misc();
    if (i >= s.size() || s[i] != '<') throw std::runtime_error("urdf: expected '<'");
    ++i;
    xml_node n;
    n.name = name();
    for (;;) {
      skip_ws();
      if (starts("/>")) { i += 2; return n; }
      if (s[i] == '>') { ++i; break; }
      std::string k = name();
      skip_ws();
      if (s[i] != '=') throw std::runtime_error("urdf: expected '='");
      ++i; skip_ws();
      const char q = s[i];
      if (q != '"' && q != '\'') throw std::runtime_error("urdf: expected a quoted value");
      size_t e = s.find(q, i + 1);
      if (e == std::string::npos) throw std::runtime_error("urdf: unterminated attribute");
      n.attr[k] = s.substr(i + 1, e - i - 1);
      i = e + 1;
    }
    for (;;) {
      skip_misc();
      if (i >= s.size()) throw std::runtime_error("urdf: unterminated element " + n.name);
      if (starts("</")) { i = s.find('>', i); if (i == std::string::npos) throw std::runtime_error("urdf: bad close tag"); ++i; return n; }
      if (s[i] == '<') n.children.push_back(element());
      else ++i;   // text content is irrelevant to URDF
    }
  }
};

inline vec3 parse_vec3(const std::string& t, vec3 dflt = vec3()) {
  if (t.empty()) return dflt;
  std::istringstream is(t);
  vec3 v;
  if (!(is >> v.v[0] >> v.v[1] >> v.v[2])) throw std::runtime_error("urdf: expected three numbers in '" + t + "'");
  return v;
}

struct link_t { double mass = 0; vec3 com; mat3 Ic, Rc; bool has_inertial = false; };
struct joint_t { std::string name, type, parent, child; vec3 xyz, axis; mat3 R; };

struct model_table {
  ddp_hip_model_storage st{};
  ddp_hip_model model{};
  std::vector<std::string> joint_names;        // of the 1-DoF joints, in table order
  std::map<std::string, std::pair<int, vec3>> frames;   // link name -> (joint index it moves with, offset of the link frame in that joint's frame)
  std::map<std::string, mat3> frame_rot;
};

// Builds the table.  Bodies rigidly attached through fixed joints are merged into the moving body they hang from (mass,
// centre of mass and inertia composed), as Pinocchio does; links fixed to the world carry no dynamics and are dropped.
inline void build(const std::string& urdf_text, bool floating_base, model_table& out) {
  xml_parser px(urdf_text);
  xml_node robot = px.element();
  if (robot.name != "robot") throw std::runtime_error("urdf: root element must be <robot>");
  std::map<std::string, link_t> links;
  std::vector<joint_t> joints;
  for (auto const& c : robot.children) {
    if (c.name == "link") {
      link_t L;
      if (const xml_node* in = c.child("inertial")) {
        L.has_inertial = true;
        if (const xml_node* o = in->child("origin")) {
          L.com = parse_vec3(o->get("xyz"));
          vec3 e = parse_vec3(o->get("rpy"));
          L.Rc = rpy(e.v[0], e.v[1], e.v[2]);
        }
        if (const xml_node* m = in->child("mass")) L.mass = std::atof(m->get("value", "0").c_str());
        if (const xml_node* I = in->child("inertia")) {
          const double ixx = std::atof(I->get("ixx", "0").c_str()), ixy = std::atof(I->get("ixy", "0").c_str()), ixz = std::atof(I->get("ixz", "0").c_str());
          const double iyy = std::atof(I->get("iyy", "0").c_str()), iyz = std::atof(I->get("iyz", "0").c_str()), izz = std::atof(I->get("izz", "0").c_str());
          mat3 Il;
          Il.m[0] = ixx; Il.m[1] = ixy; Il.m[2] = ixz; Il.m[3] = ixy; Il.m[4] = iyy; Il.m[5] = iyz; Il.m[6] = ixz; Il.m[7] = iyz; Il.m[8] = izz;
          L.Ic = mul(mul(L.Rc, Il), transpose(L.Rc));     // inertia about the com, expressed in the link frame
        }
      }
      links[c.get("name")] = L;
    } else if (c.name == "joint") {
      joint_t J;
      J.name = c.get("name"); J.type = c.get("type");
      if (const xml_node* p = c.child("parent")) J.parent = p->get("link");
      if (const xml_node* ch = c.child("child")) J.child = ch->get("link");
      if (const xml_node* o = c.child("origin")) {
        J.xyz = parse_vec3(o->get("xyz"));
        vec3 e = parse_vec3(o->get("rpy"));
        J.R = rpy(e.v[0], e.v[1], e.v[2]);
      }
      vec3 ax; ax.v[0] = 1;
      if (const xml_node* a = c.child("axis")) ax = parse_vec3(a->get("xyz"), ax);
      const double nrm = std::sqrt(ax.v[0] * ax.v[0] + ax.v[1] * ax.v[1] + ax.v[2] * ax.v[2]);
      if (nrm > 0) for (double& x : ax.v) x /= nrm;
      J.axis = ax;
      if (c.child("mimic")) throw std::runtime_error("urdf: mimic joints are not supported (" + J.name + ")");
      joints.push_back(J);
    }
  }
  // root link: the one that is nobody's child
  std::map<std::string, const joint_t*> joint_of_child;
  for (auto const& j : joints) joint_of_child[j.child] = &j;
  std::string root;
  for (auto const& kv : links) if (!joint_of_child.count(kv.first)) { if (!root.empty()) throw std::runtime_error("urdf: more than one root link"); root = kv.first; }
  if (root.empty()) throw std::runtime_error("urdf: no root link");
  std::map<std::string, std::vector<const joint_t*>> children_of;
  for (auto const& j : joints) children_of[j.parent].push_back(&j);

  // moving bodies in depth-first order (parents before children)
  struct body_t { int parent; int jtype; vec3 axis; mat3 Rp; vec3 pp; double mass = 0; vec3 mc; mat3 Io; std::string name; };
  std::vector<body_t> bodies;
  auto add_inertia = [](body_t& b, const link_t& L, const mat3& R, const vec3& p) {
    // link frame placed at (R, p) in the body frame: accumulate mass, first moment and inertia about the body origin
    if (!L.has_inertial || L.mass <= 0) return;
    vec3 c = mul(R, L.com);
    for (int k = 0; k < 3; ++k) c.v[k] += p.v[k];
    mat3 I = mul(mul(R, L.Ic), transpose(R));
    const double m = L.mass;
    const double cc = c.v[0] * c.v[0] + c.v[1] * c.v[1] + c.v[2] * c.v[2];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) b.Io.m[3 * i + j] += I.m[3 * i + j] + m * ((i == j ? cc : 0.0) - c.v[i] * c.v[j]);
    b.mass += m;
    for (int k = 0; k < 3; ++k) b.mc.v[k] += m * c.v[k];
  };
  int base = -1;
  if (floating_base) {
    // 3 prismatic (x, y, z) + 3 revolute (x, y, z), massless except the last, which carries the root link
    for (int k = 0; k < 6; ++k) {
      body_t b{}; b.parent = k - 1; b.jtype = k < 3 ? DDP_HIP_JOINT_PRISMATIC : DDP_HIP_JOINT_REVOLUTE;
      b.axis.v[k % 3] = 1; b.Io = mat3(); for (double& x : b.Io.m) x = 0;
      b.name = std::string("base_") + (k < 3 ? "p" : "r") + "xyz"[k % 3];
      bodies.push_back(b);
    }
    base = 5;
  }
  struct item_t { std::string link; int body; mat3 R; vec3 p; };   // link frame = (R, p) in frame of `body` (-1: world, fixed)
  std::vector<item_t> stack{{root, base, mat3(), vec3()}};
  if (base >= 0) { add_inertia(bodies[5], links[root], mat3(), vec3()); out.frames[root] = {5, vec3()}; out.frame_rot[root] = mat3(); }
  while (!stack.empty()) {
    item_t it = stack.back(); stack.pop_back();
    for (const joint_t* j : children_of[it.link]) {
      // joint frame in the current body's frame
      mat3 Rj = mul(it.R, j->R);
      vec3 pj = mul(it.R, j->xyz);
      for (int k = 0; k < 3; ++k) pj.v[k] += it.p.v[k];
      if (j->type == "fixed") {
        if (it.body >= 0) { add_inertia(bodies[it.body], links[j->child], Rj, pj); out.frames[j->child] = {it.body, pj}; out.frame_rot[j->child] = Rj; }
        stack.push_back({j->child, it.body, Rj, pj});
      } else if (j->type == "revolute" || j->type == "continuous" || j->type == "prismatic") {
        body_t b{};
        b.parent = it.body; b.jtype = j->type == "prismatic" ? DDP_HIP_JOINT_PRISMATIC : DDP_HIP_JOINT_REVOLUTE;
        b.axis = j->axis; b.Rp = Rj; b.pp = pj; for (double& x : b.Io.m) x = 0;
        b.name = j->name;
        bodies.push_back(b);
        const int idx = (int)bodies.size() - 1;
        add_inertia(bodies[idx], links[j->child], mat3(), vec3());
        out.frames[j->child] = {idx, vec3()}; out.frame_rot[j->child] = mat3();
        stack.push_back({j->child, idx, mat3(), vec3()});
      } else {
        throw std::runtime_error("urdf: joint type '" + j->type + "' of " + j->name + " is not supported (1-DoF and fixed joints only)");
      }
    }
  }
  // depth-first with a stack visits children after parents but not necessarily in index order: parent[i] < i must hold
  const int N = (int)bodies.size();
  if (N < 1) throw std::runtime_error("urdf: no moving joint");
  if (N > DDP_HIP_MAX_JOINTS) throw std::runtime_error("urdf: more than DDP_HIP_MAX_JOINTS joints");
  for (int i = 0; i < N; ++i) if (bodies[i].parent >= i) throw std::runtime_error("urdf: internal ordering error");
  ddp_hip_model_storage& st = out.st;
  std::memset(&st, 0, sizeof(st));
  for (int i = 0; i < N; ++i) {
    const body_t& b = bodies[i];
    st.parent[i] = b.parent; st.jtype[i] = b.jtype;
    for (int k = 0; k < 3; ++k) { st.axis[3 * i + k] = b.axis.v[k]; st.pp[3 * i + k] = b.pp.v[k]; }
    for (int k = 0; k < 9; ++k) st.Rp[9 * i + k] = b.Rp.m[k];
    st.mass_j[i] = b.mass;
    vec3 c;
    if (b.mass > 0) for (int k = 0; k < 3; ++k) c.v[k] = b.mc.v[k] / b.mass;
    for (int k = 0; k < 3; ++k) st.com[3 * i + k] = c.v[k];
    // inertia about the com from the inertia about the body origin (parallel axes)
    const double cc = c.v[0] * c.v[0] + c.v[1] * c.v[1] + c.v[2] * c.v[2];
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) st.Ic[9 * i + 3 * r + q] = b.Io.m[3 * r + q] - b.mass * ((r == q ? cc : 0.0) - c.v[r] * c.v[q]);
    out.joint_names.push_back(b.name);
  }
  ddp_hip_model& m = out.model;
  std::memset(&m, 0, sizeof(m));
  m.kind = DDP_HIP_MODEL_TREE; m.nv = N;
  m.parent = st.parent; m.jtype = st.jtype; m.axis = st.axis; m.Rp = st.Rp; m.pp = st.pp; m.mass_j = st.mass_j; m.com = st.com; m.Ic = st.Ic;
  m.gravity[0] = 0; m.gravity[1] = 0; m.gravity[2] = -9.81;     // pinocchio::Model::gravity981
}

inline void build_from_file(const std::string& path, bool floating_base, model_table& out) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("urdf: cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  build(ss.str(), floating_base, out);
}

}  // namespace ddp_hip_urdf
#endif
