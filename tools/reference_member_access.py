#!/usr/bin/env python3
"""Which access specifier do the members the adapters touch have in the reference's headers?

The adapters (adapters/*.hpp, *.ipp, *.cpp) are written against the reference's headers as they stand and cannot be compiled
in this image (Eigen / Boost / fmt are absent), so a member that is `private:` there (round 2: pendulum_model_t::m_mass)
goes unnoticed.  This script reads the reference's headers AS TEXT, walks every `struct` / `class` body, tracks
`public:` / `private:` / `protected:` at the body's own nesting depth and records where each requested member is declared.
Output: tests/golden/reference_member_access.json (data: header, struct, member, access, line -- no source text), which
tests/test_adapters.py checks the adapters against; where /root/reference is present the test regenerates the table and
compares.

usage: tools/reference_member_access.py [reference_root] > tests/golden/reference_member_access.json
"""
import json
import os
import re
import sys

# (header relative to <reference>/include/ddp, struct, member): every member of a reference type the adapters name
USED = [
    ("detail/mat_seq.hpp", "mat_seq_t", "m_data"),
    ("detail/mat_seq.hpp", "mat_seq_t", "m_idx"),
    ("detail/tensor.hpp", "tensor_seq_t", "m_data"),
    ("detail/mat_seq_common.hpp", "affine_vector_function_seq_t", "m_origin"),
    ("detail/mat_seq_common.hpp", "affine_vector_function_seq_t", "m_val_data"),
    ("detail/mat_seq_common.hpp", "affine_vector_function_seq_t", "m_jac_data"),
    ("trajectory.hpp", "trajectory_t", "m_state_data"),
    ("trajectory.hpp", "trajectory_t", "m_control_data"),
    ("pendulum_model.hpp", "pendulum_model_t", "dynamics_aba"),
    ("pendulum_model.hpp", "pendulum_model_t", "m_mass"),       # private: the bridge must NOT name it (checked the other way round)
    ("pendulum_model.hpp", "pendulum_model_t", "m_length"),
    ("problem.hpp", "dynamics_t", "m_model"),
    ("problem.hpp", "dynamics_t", "dt"),
    ("problem.hpp", "dynamics_t", "second_order_finite_diff"),
    ("problem.hpp", "constraint_advance_time_t", "m_constraint"),
    ("problem.hpp", "constraint_advance_time_t", "m_dynamics"),
    ("problem.hpp", "config_constraint_t", "m_constraint_target_view"),
    ("problem.hpp", "config_constraint_t", "m_dynamics"),
    ("problem.hpp", "spatial_constraint_t", "m_constraint_target_view"),
    ("problem.hpp", "spatial_constraint_t", "m_dynamics"),
    ("problem.hpp", "spatial_constraint_t", "m_frame_id"),
    ("problem.hpp", "problem_t", "m_dynamics"),
    ("problem.hpp", "problem_t", "m_constraint"),
    ("problem.hpp", "problem_t", "c"),
    ("ddp.hpp", "ddp_solver_t", "prob"),
    ("ddp.hpp", "ddp_solver_t", "eq_idx"),
    ("ddp.hpp", "ddp_solver_t", "u_idx"),
    ("ddp.hpp", "ddp_solver_t", "x_init"),
    ("ddp.hpp", "ddp_solver_t", "index_begin"),
    ("ddp.hpp", "ddp_solver_t", "index_end"),
    ("ddp.hpp", "ddp_solver_t", "backward_pass"),
    ("ddp.hpp", "ddp_solver_t", "forward_pass"),
    ("ddp.hpp", "derivative_storage_t", "lfx"),
    ("ddp.hpp", "derivative_storage_t", "lfxx"),
    ("ddp.hpp", "derivative_storage_t", "fxx"),
    ("ddp.hpp", "derivative_storage_t", "eq_uu"),
    ("pinocchio_model.hpp", "model_t", "m_model"),              # private (opaque impl): the B2 translation unit defines it, nobody reads it
    ("pinocchio_model.hpp", "model_t", "dynamics_aba"),
]


def strip(text):
    text = re.sub(r"/\*.*?\*/", lambda m: re.sub(r"[^\n]", " ", m.group(0)), text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r'"(?:\\.|[^"\\])*"', '""', text)


def struct_bodies(text, name):
    """(start, end) offsets of the bodies of `struct name` / `class name` definitions (not forward declarations)"""
    out = []
    for m in re.finditer(r"\b(struct|class)\s+" + re.escape(name) + r"\b[^;{]*\{", text):
        i, depth = m.end(), 1
        while depth and i < len(text):
            depth += text[i] == "{"
            depth -= text[i] == "}"
            i += 1
        out.append((m.group(1), m.end(), i - 1))
    return out


def access_of(text, name, member):
    """access specifier and line of the first declaration of `member` at the top nesting level of `struct name`"""
    for kind, a, b in struct_bodies(text, name):
        access = "public" if kind == "struct" else "private"
        depth, i = 0, a
        decl = re.compile(r"\b" + re.escape(member) + r"\b\s*(;|\{|\(|=|:|\[)")
        while i < b:
            ch = text[i]
            if ch in "{(":
                depth += 1
            elif ch in "})":
                depth -= 1
            elif depth == 0:
                m = re.match(r"(public|private|protected)\s*:", text[i:i + 12])
                if m and (i == 0 or not (text[i - 1].isalnum() or text[i - 1] == "_")):
                    access = m.group(1)
                    i += m.end()
                    continue
                m = decl.match(text, i)
                if m and not (text[i - 1].isalnum() or text[i - 1] in "_.>:"):
                    return access, text.count("\n", 0, i) + 1
            i += 1
    return None, None


def table(root):
    rows = []
    cache = {}
    for header, struct, member in USED:
        path = os.path.join(root, "include", "ddp", header)
        if path not in cache:
            cache[path] = strip(open(path).read())
        access, line = access_of(cache[path], struct, member)
        rows.append({"header": "include/ddp/" + header, "struct": struct, "member": member, "access": access, "line": line})
    return rows


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    json.dump(table(root), sys.stdout, indent=1)
    sys.stdout.write("\n")
