// host test of adapters/urdf_reader.hpp (dependency-free): a three-link arm with a fixed tool flange, checked against the
// hand-computed table.  Built and run by tests/test_adapters.py with g++.
#include <cmath>
#include <cstdio>

#include "../adapters/urdf_reader.hpp"

static const char* URDF = R"(<?xml version="1.0"?>
<!-- a small arm -->
<robot name="arm3">
  <link name="base"/>
  <link name="l1"><inertial><origin xyz="0 0 0.1" rpy="0 0 0"/><mass value="2.0"/><inertia ixx="0.01" ixy="0" ixz="0" iyy="0.02" iyz="0" izz="0.03"/></inertial></link>
  <link name="l2"><inertial><origin xyz="0.2 0 0"/><mass value="1.0"/><inertia ixx="0.001" ixy="0" ixz="0" iyy="0.002" iyz="0" izz="0.003"/></inertial></link>
  <link name="l3"><inertial><origin xyz="0 0 0"/><mass value="0.5"/><inertia ixx="0.1" ixy="0" ixz="0" iyy="0.1" iyz="0" izz="0.1"/></inertial></link>
  <link name="tool"><inertial><origin xyz="0 0 0.05"/><mass value="0.5"/><inertia ixx="0" ixy="0" ixz="0" iyy="0" iyz="0" izz="0"/></inertial></link>
  <joint name="j1" type="revolute"><parent link="base"/><child link="l1"/><origin xyz="0 0 0.3" rpy="0 0 0"/><axis xyz="0 0 1"/></joint>
  <joint name="j2" type="continuous"><parent link="l1"/><child link="l2"/><origin xyz="0 0 0.2" rpy="0 0 1.5707963267948966"/><axis xyz="0 2 0"/></joint>
  <joint name="j3" type="prismatic"><parent link="l2"/><child link="l3"/><origin xyz="0.4 0 0"/><axis xyz="1 0 0"/></joint>
  <joint name="flange" type="fixed"><parent link="l3"/><child link="tool"/><origin xyz="0 0 0.1"/></joint>
</robot>)";

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++fails; } } while (0)
static bool near(double a, double b) { return std::fabs(a - b) <= 1e-12 * (1.0 + std::fabs(b)); }

int main() {
  ddp_hip_urdf::model_table t;
  ddp_hip_urdf::build(URDF, false, t);
  CHECK(t.model.nv == 3 && t.model.kind == DDP_HIP_MODEL_TREE);
  CHECK(t.st.parent[0] == -1 && t.st.parent[1] == 0 && t.st.parent[2] == 1);
  CHECK(t.st.jtype[0] == DDP_HIP_JOINT_REVOLUTE && t.st.jtype[1] == DDP_HIP_JOINT_REVOLUTE && t.st.jtype[2] == DDP_HIP_JOINT_PRISMATIC);
  CHECK(near(t.st.axis[2], 1.0) && near(t.st.axis[3 + 1], 1.0) && near(t.st.axis[6 + 0], 1.0));   // axis (0, 2, 0) is normalised
  CHECK(near(t.st.pp[2], 0.3) && near(t.st.pp[3 + 2], 0.2) && near(t.st.pp[6 + 0], 0.4));
  // j2's frame is yawed by 90 degrees: Rp = Rz(pi/2)
  CHECK(std::fabs(t.st.Rp[9 + 0]) < 1e-15 && near(t.st.Rp[9 + 1], -1.0) && near(t.st.Rp[9 + 3], 1.0) && near(t.st.Rp[9 + 8], 1.0));
  CHECK(near(t.st.mass_j[0], 2.0) && near(t.st.mass_j[1], 1.0));
  // body 3 = l3 + the tool flange merged: mass 1.0, com at (0, 0, 0.075), inertia by the parallel-axis theorem
  CHECK(near(t.st.mass_j[2], 1.0));
  CHECK(near(t.st.com[6 + 2], 0.075) && std::fabs(t.st.com[6 + 0]) < 1e-15);
  const double izz = 0.1, ixx = 0.1 + 0.5 * 0.075 * 0.075 + 0.5 * 0.075 * 0.075;   // two point-ish masses 0.075 from the common com
  CHECK(near(t.st.Ic[18 + 8], izz) && near(t.st.Ic[18 + 0], ixx) && near(t.st.Ic[18 + 4], ixx));
  CHECK(t.frames.count("tool") && t.frames["tool"].first == 2 && near(t.frames["tool"].second.v[2], 0.1));
  CHECK(near(t.model.gravity[2], -9.81));
  // floating base: 6 extra joints in front, the root link rides on the sixth
  ddp_hip_urdf::model_table f;
  ddp_hip_urdf::build(URDF, true, f);
  CHECK(f.model.nv == 9 && f.st.jtype[0] == DDP_HIP_JOINT_PRISMATIC && f.st.jtype[5] == DDP_HIP_JOINT_REVOLUTE && f.st.parent[6] == 5);
  bool threw = false;
  try { ddp_hip_urdf::model_table b; ddp_hip_urdf::build("<robot><link name='a'/><link name='b'/><joint name='j' type='floating'><parent link='a'/><child link='b'/></joint></robot>", false, b); }
  catch (const std::exception&) { threw = true; }
  CHECK(threw);
  std::printf(fails ? "test_urdf: %d FAILED\n" : "test_urdf: ok\n", fails);
  return fails ? 1 : 0;
}
