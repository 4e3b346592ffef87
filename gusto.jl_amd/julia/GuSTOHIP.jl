# GuSTOHIP.jl -- thin `ccall` wrapper that plugs libgusto_hip.so into GuSTO.jl's own solver seam:
#
#     solve_SCP!(TOS, TOP, solve_gusto_hip!, init_traj_straightline, "hip")
#
# `solve_gusto_hip!` has the positional signature of `solve_gusto_jump!` (src/scp/scp_gusto.jl:49), which is what
# `solve_SCP!` calls through its function argument (src/traj_opt.jl:47-72).  Include this file after
# `include("src/GuSTO.jl")`.  No CUDA.jl / AMDGPU.jl: every device interaction is behind the C ABI of
# include/gusto_hip.h.  (Julia is not available in the build image: this file mirrors gusto.jl_amd/host.py, which
# IS exercised by the test-suite, call for call.)

const libgusto_hip = get(ENV, "LIBGUSTO_HIP", joinpath(@__DIR__, "..", "libgusto_hip.so"))

struct GustoScpParams      # gusto_scp_params
  Delta0::Cdouble; omega0::Cdouble; omega_max::Cdouble; eps::Cdouble; rho0::Cdouble; rho1::Cdouble
  beta_succ::Cdouble; beta_fail::Cdouble; gamma_fail::Cdouble; convergence_threshold::Cdouble
end

mutable struct GustoHistory  # gusto_history
  hist_cap::Cint
  n_hist::Ptr{Cint}; nJ::Ptr{Cint}; n_rho::Ptr{Cint}
  J_true::Ptr{Cdouble}; J_full::Ptr{Cdouble}; convergence_measure::Ptr{Cdouble}
  Delta::Ptr{Cdouble}; omega::Ptr{Cdouble}; rho::Ptr{Cdouble}
  accept_solution::Ptr{Cint}; scp_status::Ptr{Cint}; solver_status::Ptr{Cint}
  trust_region_satisfied::Ptr{Cint}; convex_ineq_satisfied::Ptr{Cint}; ipm_iters::Ptr{Cint}
end

# gusto_model_params (include/gusto_hip.h): robot + model scalars of SCPP.PD.robot / SCPP.PD.model, so that a user's
# Freeflyer(...) / model.clearance reach the kernels (robot/freeflyer.jl:28-62, robot/astrobee3D.jl:15-33, dubins_car.jl:22-33)
struct GustoModelParams
  mass::Cdouble; Jdiag::NTuple{3,Cdouble}; radius::Cdouble; clearance::Cdouble
  hard_limit_vel::Cdouble; hard_limit_accel::Cdouble; hard_limit_omega::Cdouble; hard_limit_alpha::Cdouble
  dubins_v::Cdouble; dubins_k::Cdouble; u_max::Cdouble; u_min::Cdouble
  x_max::NTuple{13,Cdouble}; x_min::NTuple{13,Cdouble}
  n_robot_comp::Cint; comp_off::NTuple{6,Cdouble}
end
gusto_pad13(v) = ntuple(i -> i <= length(v) ? Float64(v[i]) : 0.0, 13)
gusto_model_params(rb::Freeflyer, model) = GustoModelParams(rb.mass_ff, (rb.J_ff, rb.J_ff, rb.J_ff), rb.r, model.clearance,
  rb.hard_limit_vel, rb.hard_limit_accel, rb.hard_limit_ω, rb.hard_limit_α, 0., 0., 0., 0., gusto_pad13(()), gusto_pad13(()),
  2, (0., 0., 0., Float64.(rb.xb)...))                         # body + arm cylinder at xb (freeflyer.jl:53-57)
gusto_model_params(rb::Astrobee3D, model) = GustoModelParams(rb.mass, (rb.J[1,1], rb.J[2,2], rb.J[3,3]), rb.r, model.clearance,
  rb.hard_limit_vel, rb.hard_limit_accel, rb.hard_limit_ω, rb.hard_limit_α, 0., 0., 0., 0., gusto_pad13(()), gusto_pad13(()),
  1, (0., 0., 0., 0., 0., 0.))
gusto_model_params(rb::Car, model::DubinsCar) = GustoModelParams(0., (0., 0., 0.), 0., model.clearance, 0., 0., 0., 0.,
  model.v, model.k, model.u_max, model.u_min, gusto_pad13(model.x_max), gusto_pad13(model.x_min), 1, (0., 0., 0., 0., 0., 0.))

gusto_model_id(::FreeflyerSE2) = 0
gusto_model_id(::DubinsCar) = 1
gusto_model_id(::AstrobeeSE3) = 2
gusto_model_id(::AstrobeeSE3Manifold) = 3

const GUSTO_SCP_STATUS = (:NA, :OK, :InaccurateModel, :ViolatesConstraints, :TrustRegionViolated)
const GUSTO_SOLVER_STATUS = (:NA, :OPTIMAL, :ALMOST_LOCALLY_SOLVED, :FAILED)

gusto_check(rc, h, what) = rc == 0 || error("gusto_$what -> $rc: " *
    unsafe_string(ccall((:gusto_last_error, libgusto_hip), Cstring, (Ptr{Cvoid},), h)))

# keep-out set in Workspace order (types.jl:19): keepout_zones then obstacle_set; AABBs first, spheres after
function gusto_env_tables(env)
  boxes, spheres = Float64[], Float64[]
  for z in (env.keepout_zones..., env.obstacle_set...)
    if z isa HyperRectangle
      lo, hi = Float64.(minimum(z)), Float64.(maximum(z))
      append!(boxes, lo); append!(boxes, hi)
    elseif z isa HyperSphere
      append!(spheres, Float64.(origin(z))); push!(spheres, Float64(radius(z)))
    else
      error("gusto_hip: unsupported keep-out primitive $(typeof(z))")
    end
  end
  boxes, spheres
end

# goals active at tf_guess -> (lo, hi): lo == hi point row, lo < hi box rows, +-Inf no goal (dynamics.jl:30-42)
function gusto_goal_bounds(goal_set, x_dim, tf_guess)
  lo, hi = fill(-Inf, x_dim), fill(Inf, x_dim)
  for goal in values(inclusive(goal_set.goals, searchsortedfirst(goal_set.goals, tf_guess), searchsortedlast(goal_set.goals, tf_guess)))
    if goal.params isa PointGoal
      lo[goal.ind_coordinates] = goal.params.point; hi[goal.ind_coordinates] = goal.params.point
    else
      lo[goal.ind_coordinates] = goal.params.lower_bound; hi[goal.ind_coordinates] = goal.params.upper_bound
    end
  end
  lo, hi
end

# every call adds its iterations plus one leading J_true / rho entry: room for a few resumed calls
gusto_hist_cap(max_iter) = max(64, 4max_iter + 16)

const GUSTO_HANDLES = IdDict{SCPSolution,Ptr{Cvoid}}()   # device-side state per solution: resume (scp_gusto.jl:67)

# gusto_get_history for a handle holding B problems: every vector as a (hist_cap + 1) x B matrix, column b = problem b
# (one spare row: after a failed subproblem solver_status has one entry more than the other vectors)
function gusto_histories(h, B)
  capref = Ref{Cint}(0)    # the capacity fixed when the handle was created, NOT a function of this call's max_iter
  gusto_check(ccall((:gusto_get_hist_cap, libgusto_hip), Cint, (Ptr{Cvoid}, Ref{Cint}), h, capref), h, "get_hist_cap")
  cap = Int(capref[])
  rows = cap + 1
  d() = zeros(Cdouble, rows, B); i() = zeros(Cint, rows, B)
  nh, nJ, nr = zeros(Cint, B), zeros(Cint, B), zeros(Cint, B)
  Jt, Jf, cm, Dv, wv, rv = d(), d(), d(), d(), d(), d()
  acc, scp, sol, trs, cvx, ipi = i(), i(), i(), i(), i(), i()
  hist = GustoHistory(rows, pointer(nh), pointer(nJ), pointer(nr), pointer(Jt), pointer(Jf), pointer(cm), pointer(Dv),
                      pointer(wv), pointer(rv), pointer(acc), pointer(scp), pointer(sol), pointer(trs), pointer(cvx), pointer(ipi))
  GC.@preserve nh nJ nr Jt Jf cm Dv wv rv acc scp sol trs cvx ipi begin
    gusto_check(ccall((:gusto_get_history, libgusto_hip), Cint, (Ptr{Cvoid}, Ref{GustoHistory}), h, hist), h, "get_history")
  end
  (cap=cap, nh=nh, nJ=nJ, nr=nr, Jt=Jt, Jf=Jf, cm=cm, Dv=Dv, wv=wv, rv=rv, acc=acc, scp=scp, sol=sol, trs=trs, cvx=cvx)
end

# SCPSolution / SCPParam_GuSTO vectors of problem b of a handle (types.jl:150-173, scp_gusto.jl:15-19) from its histories
function gusto_fill_solution!(SCPS, alg, Hs, b, its, conv, succ, stop, dual)
  H, J, R = Hs.nh[b], Hs.nJ[b], Hs.nr[b]
  stop == 4 && error("gusto_hip: history capacity ($(Hs.cap)) reached before iter_cap; release the handle " *
                     "(gusto_release!) and solve again with a larger max_iter (or hist_cap) on the first call")
  SCPS.J_true, SCPS.J_full = Hs.Jt[1:J, b], Hs.Jf[1:J, b]
  # a failed subproblem pushes its status and nothing else before the early return (scp_gusto.jl:106-111)
  SCPS.solver_status = [GUSTO_SOLVER_STATUS[s+1] for s in Hs.sol[1:(stop == 2 ? H + 1 : H), b]]
  SCPS.scp_status = [GUSTO_SCP_STATUS[s+1] for s in Hs.scp[1:H, b]]
  SCPS.accept_solution = Bool.(Hs.acc[1:H, b])
  SCPS.convergence_measure = Hs.cm[1:H, b]
  SCPS.iterations, SCPS.converged, SCPS.successful = its, conv != 0, succ != 0
  SCPS.dual = dual
  alg.Δ_vec, alg.ω_vec, alg.ρ_vec = Hs.Dv[1:H, b], Hs.wv[1:H, b], Hs.rv[1:R, b]
  alg.trust_region_satisfied_vec, alg.convex_ineq_satisfied_vec = Bool.(Hs.trs[1:H, b]), Bool.(Hs.cvx[1:H, b])
  stop == 3 && @warn "GuSTO SCP omegamax exceeded"     # scp_gusto.jl:163-166 (`warn` is undefined on Julia >= 0.7)
  nothing
end

# `hist_cap`: capacity of the history vectors of the handle the FIRST call creates; a caller that resumes in short calls
# (solve_SCPshooting!: one iteration per call, two entries each) passes gusto_hist_cap of its whole budget
function solve_gusto_hip!(SCPS::SCPSolution, SCPP::SCPProblem, solver="hip", max_iter=30, force=false; device=0, hist_cap=0, kwarg...)
  model, N = SCPP.PD.model, SCPP.N
  n, m = model.x_dim, model.u_dim
  !isdefined(SCPP.param, :alg) ? SCPP.param.alg = SCPParam_GuSTO(model) : nothing
  alg = SCPP.param.alg
  h = get(GUSTO_HANDLES, SCPS, C_NULL)
  if h == C_NULL
    href = Ref{Ptr{Cvoid}}(C_NULL)
    gusto_check(ccall((:gusto_create, libgusto_hip), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Cint),
                      href, gusto_model_id(model), N, 1, hist_cap > 0 ? hist_cap : gusto_hist_cap(max_iter), device), href[], "create")
    h = href[]
    sp = GustoScpParams(alg.Δ0, alg.ω0, alg.ω_max, alg.ε, alg.ρ0, alg.ρ1, alg.β_succ, alg.β_fail, alg.γ_fail,
                        SCPP.param.convergence_threshold)
    gusto_check(ccall((:gusto_set_params, libgusto_hip), Cint, (Ptr{Cvoid}, Ref{GustoScpParams}, Ref{GustoModelParams}),
                      h, sp, gusto_model_params(SCPP.PD.robot, model)), h, "set_params")
    boxes, spheres = gusto_env_tables(SCPP.PD.env)
    gusto_check(ccall((:gusto_set_env, libgusto_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}),
                      h, length(boxes) ÷ 6, boxes, length(spheres) ÷ 4, spheres), h, "set_env")
    lo, hi = gusto_goal_bounds(SCPP.PD.goal_set, n, SCPP.tf_guess)
    X0, U0 = Matrix{Float64}(SCPS.traj.X), Matrix{Float64}(SCPS.traj.U)   # Julia column-major X[n,N] == C X[k][i]
    gusto_check(ccall((:gusto_set_problems, libgusto_hip), Cint,
                      (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                      h, 1, Float64.(SCPP.PD.x_init), lo, hi, [Float64(SCPP.tf_guess)], X0, U0), h, "set_problems")
    GUSTO_HANDLES[SCPS] = h
  end
  t0 = time_ns()
  gusto_check(ccall((:gusto_solve, libgusto_hip), Cint, (Ptr{Cvoid}, Cint, Cint), h, max_iter, force), h, "solve")
  elapsed = (time_ns() - t0) / 10^9

  X, U = zeros(n, N), zeros(m, N)
  gusto_check(ccall((:gusto_get_traj, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), h, X, U), h, "get_traj")
  SCPS.traj.X, SCPS.traj.U = X, U                      # TOS.traj aliases SCPS.traj (traj_opt.jl:58)
  its, conv, succ, stop, ipm = (zeros(Cint, 1) for _ in 1:5)
  gusto_check(ccall((:gusto_get_status, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}),
                    h, its, conv, succ, stop, ipm), h, "get_status")
  dual = zeros(n)
  gusto_check(ccall((:gusto_get_dual, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), h, dual), h, "get_dual")
  gusto_fill_solution!(SCPS, alg, gusto_histories(h, 1), 1, its[1], conv[1], succ[1], stop[1], dual)
  SCPS.total_time += elapsed
  SCPS.iter_elapsed_times = vcat(0., fill(SCPS.total_time / max(1, its[1]), its[1]))
  SCPP.param.obstacle_toggle_distance = alg.Δ_vec[end] / 8 + model.clearance
  nothing
end

gusto_release!(SCPS::SCPSolution) = (h = pop!(GUSTO_HANDLES, SCPS, C_NULL); h != C_NULL && ccall((:gusto_destroy, libgusto_hip), Cint, (Ptr{Cvoid},), h); nothing)
