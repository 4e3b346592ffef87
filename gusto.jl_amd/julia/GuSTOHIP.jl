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

function solve_gusto_hip!(SCPS::SCPSolution, SCPP::SCPProblem, solver="hip", max_iter=30, force=false; device=0, kwarg...)
  model, N = SCPP.PD.model, SCPP.N
  n, m = model.x_dim, model.u_dim
  !isdefined(SCPP.param, :alg) ? SCPP.param.alg = SCPParam_GuSTO(model) : nothing
  alg = SCPP.param.alg
  h = get(GUSTO_HANDLES, SCPS, C_NULL)
  if h == C_NULL
    href = Ref{Ptr{Cvoid}}(C_NULL)
    gusto_check(ccall((:gusto_create, libgusto_hip), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Cint),
                      href, gusto_model_id(model), N, 1, gusto_hist_cap(max_iter), device), href[], "create")
    h = href[]
    sp = GustoScpParams(alg.Δ0, alg.ω0, alg.ω_max, alg.ε, alg.ρ0, alg.ρ1, alg.β_succ, alg.β_fail, alg.γ_fail,
                        SCPP.param.convergence_threshold)
    gusto_check(ccall((:gusto_set_params, libgusto_hip), Cint, (Ptr{Cvoid}, Ref{GustoScpParams}, Ptr{Cvoid}), h, sp, C_NULL), h, "set_params")
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
  capref = Ref{Cint}(0)    # the capacity fixed when the handle was created, NOT a function of this call's max_iter
  gusto_check(ccall((:gusto_get_hist_cap, libgusto_hip), Cint, (Ptr{Cvoid}, Ref{Cint}), h, capref), h, "get_hist_cap")
  cap = Int(capref[])
  d() = zeros(Cdouble, cap + 1); i() = zeros(Cint, cap + 1)
  nh, nJ, nr = zeros(Cint, 1), zeros(Cint, 1), zeros(Cint, 1)
  Jt, Jf, cm, Dv, wv, rv = d(), d(), d(), d(), d(), d()
  acc, scp, sol, trs, cvx, ipi = i(), i(), i(), i(), i(), i()
  hist = GustoHistory(cap, pointer(nh), pointer(nJ), pointer(nr), pointer(Jt), pointer(Jf), pointer(cm), pointer(Dv),
                      pointer(wv), pointer(rv), pointer(acc), pointer(scp), pointer(sol), pointer(trs), pointer(cvx), pointer(ipi))
  GC.@preserve nh nJ nr Jt Jf cm Dv wv rv acc scp sol trs cvx ipi begin
    gusto_check(ccall((:gusto_get_history, libgusto_hip), Cint, (Ptr{Cvoid}, Ref{GustoHistory}), h, hist), h, "get_history")
  end
  H, J, R = nh[1], nJ[1], nr[1]
  stop[1] == 4 && error("gusto_hip: history capacity ($cap) reached before iter_cap; release the handle " *
                        "(gusto_release!) and solve again with a larger max_iter on the first call")
  SCPS.J_true, SCPS.J_full = Jt[1:J], Jf[1:J]
  # a failed subproblem pushes its status and nothing else before the early return (scp_gusto.jl:106-111)
  SCPS.solver_status = [GUSTO_SOLVER_STATUS[s+1] for s in sol[1:(stop[1] == 2 ? H + 1 : H)]]
  SCPS.scp_status = [GUSTO_SCP_STATUS[s+1] for s in scp[1:H]]
  SCPS.accept_solution = Bool.(acc[1:H])
  SCPS.convergence_measure = cm[1:H]
  SCPS.iterations, SCPS.converged, SCPS.successful = its[1], conv[1] != 0, succ[1] != 0
  SCPS.total_time += elapsed
  SCPS.iter_elapsed_times = vcat(0., fill(SCPS.total_time / max(1, its[1]), its[1]))
  dual = zeros(n)
  gusto_check(ccall((:gusto_get_dual, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), h, dual), h, "get_dual")
  SCPS.dual = dual
  alg.Δ_vec, alg.ω_vec, alg.ρ_vec = Dv[1:H], wv[1:H], rv[1:R]
  alg.trust_region_satisfied_vec, alg.convex_ineq_satisfied_vec = Bool.(trs[1:H]), Bool.(cvx[1:H])
  SCPP.param.obstacle_toggle_distance = alg.Δ_vec[end] / 8 + model.clearance
  stop[1] == 3 && @warn "GuSTO SCP omegamax exceeded"     # scp_gusto.jl:163-166 (`warn` is undefined on Julia >= 0.7)
  nothing
end

# solve!(SS, SP) on the GPU (src/shooting.jl:4-49; DubinsCar and AstrobeeSE3Manifold): the handle that holds the SCP state of SP's problem runs the
# batched indirect shooting from SP.p0 (= SCPS.dual).  Use it in place of `solve!` inside solve_SCPshooting!
# (src/traj_opt.jl:28): `ss_sol = solve_shooting_hip!(SS, SP, SCPS)`.
struct GustoShootOpts; substeps::Cint; max_newton::Cint; ftol::Cdouble; end
function solve_shooting_hip!(SS::ShootingSolution, SP::ShootingProblem, SCPS::SCPSolution; substeps=4, max_newton=100, ftol=1e-3)
  h = get(GUSTO_HANDLES, SCPS, C_NULL)
  h == C_NULL && error("solve_shooting_hip!: run solve_gusto_hip! on this SCPSolution first")
  model, N = SP.PD.model, SP.N
  n, m = model.x_dim, model.u_dim
  t0 = time_ns()
  gusto_check(ccall((:gusto_shoot, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ref{GustoShootOpts}),
                    h, Float64.(SP.p0), GustoShootOpts(substeps, max_newton, ftol)), h, "shoot")
  st, it, res, p0 = zeros(Cint, 1), zeros(Cint, 1), zeros(1), zeros(n)
  X, U = zeros(n, N), zeros(m, N)
  gusto_check(ccall((:gusto_get_shoot, libgusto_hip), Cint,
                    (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), h, st, it, res, p0, X, U), h, "get_shoot")
  el = (time_ns() - t0) / 10^9
  if st[1] == 1                                   # sol_newton.f_converged
    new_traj = Trajectory(X, U, SP.tf)
    push!(SS.prob_status, :Optimal); push!(SS.J_true, cost_true(new_traj, new_traj, SP))
    push!(SS.convergence_measure, convergence_metric(new_traj, SS.traj, SP)); copy!(SS.traj, new_traj)
  else
    push!(SS.prob_status, :Diverged); push!(SS.J_true, NaN); push!(SS.convergence_measure, NaN)
  end
  push!(SS.iter_elapsed_times, el)
  nothing
end

gusto_release!(SCPS::SCPSolution) = (h = pop!(GUSTO_HANDLES, SCPS, C_NULL); h != C_NULL && ccall((:gusto_destroy, libgusto_hip), Cint, (Ptr{Cvoid},), h); nothing)

# Batch entry point (the reference has none): every TOP must share model, N and environment.  `devices` = GPU ordinals:
# the problems are split in contiguous blocks of ceil(B/G) (SURVEY.md 8(e)), one handle per entry, every block enqueued
# with gusto_solve_async so the GPUs run concurrently; the results come back in problem order.
function solve_SCP_batch!(TOSs::Vector, TOPs::Vector, init_method=init_traj_straightline; max_iter=30, force=false, device=0, devices=nothing)
  TOP0 = TOPs[1]; model, N = TOP0.PD.model, TOP0.N
  n, m, B = model.x_dim, model.u_dim, length(TOPs)
  all(T -> typeof(T.PD.model) == typeof(model) && T.N == N && T.PD.env === TOP0.PD.env, TOPs) ||
    error("solve_SCP_batch!: all problems must share the model type, N and the environment")
  devs = devices === nothing ? [device] : collect(devices)
  G = length(devs); per = cld(B, G)
  boxes, spheres = gusto_env_tables(TOP0.PD.env)
  x0 = hcat((Float64.(T.PD.x_init) for T in TOPs)...)
  bounds = [gusto_goal_bounds(T.PD.goal_set, n, T.tf_guess) for T in TOPs]
  lo, hi = hcat(first.(bounds)...), hcat(last.(bounds)...)
  tf = Float64[T.tf_guess for T in TOPs]
  inits = [init_method(T) for T in TOPs]
  X0, U0 = cat((t.X for t in inits)..., dims=3), cat((t.U for t in inits)..., dims=3)   # [n,N,B]: problem slowest
  shards = Tuple{Int,Int,Ptr{Cvoid}}[]
  for (r, dv) in enumerate(devs)
    b0, b1 = min(B, (r - 1) * per) + 1, min(B, r * per)
    b1 < b0 && continue
    href = Ref{Ptr{Cvoid}}(C_NULL)
    gusto_check(ccall((:gusto_create, libgusto_hip), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Cint),
                      href, gusto_model_id(model), N, b1 - b0 + 1, gusto_hist_cap(max_iter), dv), href[], "create")
    h = href[]
    gusto_check(ccall((:gusto_set_env, libgusto_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}),
                      h, length(boxes) ÷ 6, boxes, length(spheres) ÷ 4, spheres), h, "set_env")
    gusto_check(ccall((:gusto_set_problems, libgusto_hip), Cint,
                      (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                      h, b1 - b0 + 1, x0[:, b0:b1], lo[:, b0:b1], hi[:, b0:b1], tf[b0:b1], X0[:, :, b0:b1], U0[:, :, b0:b1]), h, "set_problems")
    gusto_check(ccall((:gusto_solve_async, libgusto_hip), Cint, (Ptr{Cvoid}, Cint, Cint), h, max_iter, force), h, "solve_async")
    push!(shards, (b0, b1, h))
  end
  for (b0, b1, h) in shards
    Bs = b1 - b0 + 1
    gusto_check(ccall((:gusto_wait, libgusto_hip), Cint, (Ptr{Cvoid},), h), h, "wait")
    X, U = zeros(n, N, Bs), zeros(m, N, Bs)
    gusto_check(ccall((:gusto_get_traj, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), h, X, U), h, "get_traj")
    its, conv, succ = zeros(Cint, Bs), zeros(Cint, Bs), zeros(Cint, Bs)
    gusto_check(ccall((:gusto_get_status, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cvoid}, Ptr{Cvoid}),
                      h, its, conv, succ, C_NULL, C_NULL), h, "get_status")
    for b in b0:b1
      SCPP = SCPProblem(TOPs[b])
      SCPS = SCPSolution(SCPP, Trajectory(X[:, :, b-b0+1], U[:, :, b-b0+1], TOPs[b].tf_guess))
      SCPS.iterations, SCPS.converged, SCPS.successful = its[b-b0+1], conv[b-b0+1] != 0, succ[b-b0+1] != 0
      TOSs[b].traj, TOSs[b].SCPS = SCPS.traj, SCPS
    end
    ccall((:gusto_destroy, libgusto_hip), Cint, (Ptr{Cvoid},), h)
  end
  nothing
end
