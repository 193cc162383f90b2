#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — applies the drop-in recipe (one library call per level operator of the reference) to the reference itself.

Reads the reference's sources where they lie (REF/src), and writes into a scratch directory copies in which the
bodies of the level operators are replaced by calls into libmgx.so; everything else — type(mg_grid)'s fields, the
cycle control of m_gs_CDS.F90, the parser m_mg_io.F90, the writers, the `mg` program — is the reference's own text.
oracle/Makefile (target `dropin`) compiles the scratch copies with amdflang, links them against the library and
leaves only the executables oracle/_ref/mg_dropin_{f32,f64}; the scratch directory is removed.

What is replaced (the callee side of the boundary; the reference's callers stay as they are):
  t_mg.f90         type mg_grid        + `type(c_ptr) :: h` (the resident level behind the grid)
                   grid_setup          -> mgx_level_set_params / _set_boxes / _setup (+ a download when grid%V exists)
                   grid_bring_back     + mgx_level_create on first use, mgx_level_bring_back
                   grid_hold_back      + mgx_level_release
                   grid_sum            -> mgx_level_abs_sum
                   grid_non_constant_elem -> mgx_level_count_nonconst
                   grid_add_box        its loop over all cells (t_mg.f90:363-390) -> mgx_box_place
  t_mg_interp.f90  grid_restriction    -> mgx_restrict (includes the child's grid_setup)
                   grid_prolongation   -> mgx_prolong (includes the parent's grid_setup)
  m_gs_CDS.F90     gs                  -> mgx_level_sweep + mgx_level_result
                   mg_gs_cds           + a download of the top level's potential at its end (mg_save reads grid%V)

usage: dropin_patch.py REF_SRC OUT_DIR [--double]
"""
import os
import re
import sys


def replace_routine(text, kind, name, new_body, keep_decl_until):
    """replace the body of `kind name(...)` ... `end kind name`; the lines up to and including the first line that
    matches keep_decl_until (the dummy-argument declarations) are kept"""
    m = re.search(rf"^[ \t]*(?:recursive[ \t]+|pure[ \t]+)?{kind}[ \t]+{name}\b.*$", text, flags=re.M | re.I)
    if not m:
        raise SystemExit(f"dropin_patch: {kind} {name} not found")
    e = re.search(rf"^[ \t]*end[ \t]+{kind}[ \t]+{name}\b.*$", text[m.end():], flags=re.M | re.I)
    if not e:
        raise SystemExit(f"dropin_patch: end of {kind} {name} not found")
    body = text[m.end():m.end() + e.start()]
    k = None
    for k in re.finditer(keep_decl_until, body, flags=re.M):
        pass
    if k is None:
        raise SystemExit(f"dropin_patch: declarations of {name} not recognised")
    return text[:m.end()] + body[:k.end()] + "\n" + new_body + "\n" + text[m.end() + e.start():]


HELPERS = """
  ! ---- added by oracle/dropin_patch.py: the level behind a grid ----
  subroutine mgx_ensure(grid)
    type(mg_grid), intent(inout) :: grid
    integer(c_int32_t) :: n(3)
    if ( .not. c_associated(mgx_the_ctx) ) call mgx_check(mgx_ctx_create(0_c_int, c_null_ptr, mgx_the_ctx))
    if ( c_associated(grid%h) ) return
    n = int(grid%n, c_int32_t)
    call mgx_check(mgx_level_create(mgx_the_ctx, int(grid%layer, c_int), n, int(storage_size(1._grid_p) / 8, c_int), grid%h))
  end subroutine mgx_ensure

  subroutine mgx_push(grid)
    type(mg_grid), intent(inout) :: grid
    real(c_double) :: a(3)
    integer(c_int32_t) :: bc(6)
    integer(c_int32_t), allocatable :: place(:), cons(:)
    real(c_double), allocatable :: val(:), rho(:)
    integer :: i
    call mgx_ensure(grid)
    a = real(grid%a, c_double)
    bc = int([grid%BC(1,1)%method, grid%BC(2,1)%method, grid%BC(1,2)%method, grid%BC(2,2)%method, &
         grid%BC(1,3)%method, grid%BC(2,3)%method], c_int32_t)
    call mgx_check(mgx_level_set_params(grid%h, real(grid%sor, c_double), a, bc))
    allocate(place(6*max(1,grid%N_box)), cons(max(1,grid%N_box)), val(max(1,grid%N_box)), rho(max(1,grid%N_box)))
    do i = 1 , grid%N_box
       place(6*i-5:6*i) = int(reshape(grid%box(i)%place, [6]), c_int32_t)
       val(i) = real(grid%box(i)%val, c_double)
       rho(i) = real(grid%box(i)%rho, c_double)
       cons(i) = merge(1_c_int32_t, 0_c_int32_t, grid%box(i)%constant)
    end do
    call mgx_check(mgx_level_set_boxes(grid%h, int(grid%N_box, c_int32_t), place, val, rho, cons))
  end subroutine mgx_push
"""


def patch_t_mg(src, double):
    s = src
    if double:
        s = s.replace("grid_p = selected_real_kind(p=6)", "grid_p = selected_real_kind(p=15)", 1)
    s = s.replace("  use t_BC\n", "  use t_BC\n  use mgx_c\n  use, intrinsic :: iso_c_binding\n", 1)
    s = s.replace("     type(mg_grid), pointer :: child => null()\n",
                  "     type(mg_grid), pointer :: child => null()\n     type(c_ptr) :: h = c_null_ptr ! the level resident on the GPU\n", 1)
    s = s.replace("\ncontains\n", "\n  type(c_ptr), save :: mgx_the_ctx = c_null_ptr\n\ncontains\n" + HELPERS, 1)
    s = replace_routine(s, "subroutine", "grid_setup", """
    integer(c_int) :: linit
    linit = 0
    if ( present(init) ) linit = merge(1, 0, init)
    call mgx_push(grid)
    call mgx_check(mgx_level_setup(grid%h, linit))
    if ( associated(grid%V) ) call mgx_check(mgx_level_download(grid%h, c_loc(grid%V)))
""", r"^.*logical, intent\(in\), optional :: init.*$")
    s = replace_routine(s, "subroutine", "grid_hold_back", """
    if ( associated(grid%V)   ) deallocate(grid%V)
    if ( associated(grid%g)   ) deallocate(grid%g)
    if ( associated(grid%g_s) ) deallocate(grid%g_s)
    nullify(grid%V,grid%g,grid%g_s)
    if ( c_associated(grid%h) ) call mgx_check(mgx_level_release(grid%h))
""", r"^.*type\(mg_grid\), intent\(inout\) :: grid.*$")
    s = replace_routine(s, "subroutine", "grid_bring_back", """
    call grid_hold_back(grid)
    allocate(grid%V(0:grid%n(1)+1,0:grid%n(2)+1,0:grid%n(3)+1))
    call mgx_ensure(grid)
    call mgx_check(mgx_level_bring_back(grid%h))
""", r"^.*type\(mg_grid\), intent\(inout\) :: grid.*$")
    s = replace_routine(s, "function", "grid_sum", """
    real(c_double) :: s
    call mgx_check(mgx_level_abs_sum(grid%h, s))
    sum = real(s, grid_p)
""", r"^.*real\(grid_p\) :: sum.*$")
    s = replace_routine(s, "function", "grid_non_constant_elem", """
    integer(c_int64_t) :: cnt
    call mgx_check(mgx_level_count_nonconst(grid%h, cnt))
    elem = int(cnt)
""", r"^.*integer :: x,y,z,elem.*$")
    # grid_add_box: the OpenMP loop over every cell becomes one library call; the rest of the routine (slot search,
    # val/rho/constant, the "at least one point" step, the recursion over the children) stays the reference's text
    m = re.search(r"^!\$OMP parallel do default\(shared\), private\(x,y,z,xyz,dyz,dz\)\n.*?^!\$OMP end parallel do\n", s, flags=re.M | re.S)
    if not m or "insert_point(box%place,x,y,z)" not in m.group(0):
        raise SystemExit("dropin_patch: the cell loop of grid_add_box not recognised")
    s = s[:m.start()] + """    if ( .not. c_associated(mgx_the_ctx) ) call mgx_check(mgx_ctx_create(0_c_int, c_null_ptr, mgx_the_ctx))
    call mgx_check(mgx_box_place(mgx_the_ctx, int(grid%n, c_int32_t), grid%dL, grid%offset, llc, box_cell, mgx_place))
    box%place = mgx_place
""" + s[m.end():]
    k = s.index("    type(mg_box), pointer :: box\n", s.index("recursive subroutine grid_add_box"))
    s = s[:k] + "    integer(c_int32_t) :: mgx_place(2,3)\n" + s[k:]
    return s


def patch_interp(src):
    s = replace_routine(src, "subroutine", "grid_restriction", """
    integer :: m
    m = grid%RES_method
    if ( m /= MG_INTERP_HALF ) m = MG_INTERP_FULL
    call mgx_push(grid%child)
    call mgx_check(mgx_restrict(grid%h, grid%child%h, int(m, c_int)))
""", r"^.*type\(mg_grid\), intent\(inout\) :: grid.*$")
    s = replace_routine(s, "subroutine", "grid_prolongation", """
    integer :: m
    m = grid%PRO_method
    if ( m /= MG_INTERP_HALF ) m = MG_INTERP_FULL
    call mgx_push(grid%parent)
    call mgx_check(mgx_prolong(grid%h, grid%parent%h, int(m, c_int)))
""", r"^.*type\(mg_grid\), intent\(inout\) :: grid.*$")
    return s


def patch_cds(src):
    s = replace_routine(src, "subroutine", "gs", """
    real(c_double) :: e2, s
    call mgx_check(mgx_level_sweep(grid%h, 1_c_int, MGX_RED_ERR))
    call mgx_check(mgx_level_result(grid%h, e2, s, c_null_ptr))
    grid%err = max(grid%err, real(e2, grid_p))
""", r"^.*type\(mg_grid\), intent\(inout\) :: grid.*$")
    m = re.search(r"^[ \t]*end[ \t]+subroutine[ \t]+mg_gs_cds\b", s, flags=re.M | re.I)
    if not m:
        raise SystemExit("dropin_patch: end of mg_gs_cds not found")
    s = s[:m.start()] + "    ! added by oracle/dropin_patch.py: mg_save reads grid%V\n" \
        "    if ( associated(top_grid%V) ) call mgx_check(mgx_level_download(top_grid%h, c_loc(top_grid%V)))\n\n" + s[m.start():]
    return s


def main():
    ref, out = sys.argv[1], sys.argv[2]
    double = "--double" in sys.argv[3:]
    here = os.path.dirname(os.path.abspath(__file__))
    os.makedirs(out, exist_ok=True)
    rd = lambda f: open(os.path.join(ref, f)).read()
    wr = lambda f, t: open(os.path.join(out, f), "w").write(t)
    wr("t_mg.f90", patch_t_mg(rd("t_mg.f90"), double))
    wr("t_mg_interp.f90", patch_interp(rd("t_mg_interp.f90")))
    wr("m_gs_CDS.F90", patch_cds(rd("m_gs_CDS.F90")))
    wr("mgx_c.f90", open(os.path.join(here, "..", "multigrid_amd", "fortran", "mgx_c.f90")).read())


if __name__ == "__main__":
    main()
