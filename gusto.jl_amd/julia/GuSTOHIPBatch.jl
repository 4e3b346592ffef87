# GuSTOHIPBatch.jl -- include after GuSTOHIP.jl: the entry points that have no counterpart with the same signature in the
# reference -- indirect shooting on the GPU for solve_SCPshooting! and the batched / multi-GPU solve.

# solve!(SS, SP) on the GPU (src/shooting.jl:4-49; DubinsCar and AstrobeeSE3Manifold): the handle that holds the SCP state of SP's problem runs the
# batched indirect shooting from SP.p0 (= SCPS.dual).  Use it in place of `solve!` inside solve_SCPshooting!
# (src/traj_opt.jl:28): `ss_sol = solve_shooting_hip!(SS, SP, SCPS)`.
struct GustoShootOpts      # gusto_shoot_opts
  substeps::Cint; max_newton::Cint; ftol::Cdouble; no_group_pass::Cint
end
function solve_shooting_hip!(SS::ShootingSolution, SP::ShootingProblem, SCPS::SCPSolution; substeps=4, max_newton=100, ftol=1e-3)
  h = get(GUSTO_HANDLES, SCPS, C_NULL)
  h == C_NULL && error("solve_shooting_hip!: run solve_gusto_hip! on this SCPSolution first")
  model, N = SP.PD.model, SP.N
  n, m = model.x_dim, model.u_dim
  t0 = time_ns()
  gusto_check(ccall((:gusto_shoot, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ref{GustoShootOpts}),
                    h, Float64.(SP.p0), GustoShootOpts(substeps, max_newton, ftol, 0)), h, "shoot")
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

# Batch entry point (the reference has none): every TOP must share model and N (each may bring its own environment).  `devices` = GPU ordinals:
# the problems are split in contiguous blocks of ceil(B/G) (SURVEY.md 8(e)), one handle per entry, every block enqueued
# with gusto_solve_async so the GPUs run concurrently; the results come back in problem order.
# gusto_set_decomposition (include/gusto_hip.h): how a solve maps problems to the GPU.  AUTO picks by batch size -- for astrobeeSE3 /
# astrobeeSE3manifold two or four wavefronts per problem (a wave per Riccati chain of the horizon) while the batch leaves SIMDs idle.
const GUSTO_DECOMP_AUTO, GUSTO_DECOMP_WAVE, GUSTO_DECOMP_LANE, GUSTO_DECOMP_WAVE2, GUSTO_DECOMP_WAVE4 = Cint(0), Cint(1), Cint(2), Cint(3), Cint(4)

function solve_SCP_batch!(TOSs::Vector, TOPs::Vector, init_method=init_traj_straightline; max_iter=30, force=false, device=0, devices=nothing,
                          decomposition=GUSTO_DECOMP_AUTO)
  TOP0 = TOPs[1]; model, N = TOP0.PD.model, TOP0.N
  n, m, B = model.x_dim, model.u_dim, length(TOPs)
  all(T -> typeof(T.PD.model) == typeof(model) && T.N == N, TOPs) ||
    error("solve_SCP_batch!: all problems must share the model type and N")
  # every ProblemDefinition owns its env (types.jl:32-39): problems with different environments go through
  # gusto_set_env_batch (one Workspace per problem), a batch sharing one env through gusto_set_env
  same_env = all(T -> T.PD.env === TOP0.PD.env, TOPs)
  envs = same_env ? nothing : [gusto_env_tables(T.PD.env) for T in TOPs]
  devs = devices === nothing ? [device] : collect(devices)
  G = length(devs); per = cld(B, G)
  boxes, spheres = gusto_env_tables(TOP0.PD.env)
  alg0 = SCPParam_GuSTO(model)
  sp = GustoScpParams(alg0.Δ0, alg0.ω0, alg0.ω_max, alg0.ε, alg0.ρ0, alg0.ρ1, alg0.β_succ, alg0.β_fail, alg0.γ_fail,
                      SCPParam(model, TOP0.fixed_final_time).convergence_threshold)
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
    decomposition == GUSTO_DECOMP_AUTO ||
      gusto_check(ccall((:gusto_set_decomposition, libgusto_hip), Cint, (Ptr{Cvoid}, Cint), h, decomposition), h, "set_decomposition")
    gusto_check(ccall((:gusto_set_params, libgusto_hip), Cint, (Ptr{Cvoid}, Ref{GustoScpParams}, Ref{GustoModelParams}),
                      h, sp, gusto_model_params(TOP0.PD.robot, model)), h, "set_params")
    gusto_check(ccall((:gusto_set_env, libgusto_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}),
                      h, length(boxes) ÷ 6, boxes, length(spheres) ÷ 4, spheres), h, "set_env")
    if !same_env
      nb = Cint[length(e[1]) ÷ 6 for e in envs[b0:b1]]; ns = Cint[length(e[2]) ÷ 4 for e in envs[b0:b1]]
      bx = vcat((vec(e[1]) for e in envs[b0:b1])...); sx = vcat((vec(e[2]) for e in envs[b0:b1])...)
      gusto_check(ccall((:gusto_set_env_batch, libgusto_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cdouble}),
                        h, b1 - b0 + 1, nb, bx, ns, sx), h, "set_env_batch")
    end
    gusto_check(ccall((:gusto_set_problems, libgusto_hip), Cint,
                      (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                      h, b1 - b0 + 1, x0[:, b0:b1], lo[:, b0:b1], hi[:, b0:b1], tf[b0:b1], X0[:, :, b0:b1], U0[:, :, b0:b1]), h, "set_problems")
    gusto_check(ccall((:gusto_solve_async, libgusto_hip), Cint, (Ptr{Cvoid}, Cint, Cint), h, max_iter, force), h, "solve_async")
    push!(shards, (b0, b1, h))
  end
  # the final gather of the multi-GPU path (north_star: "RCCL over xGMI for the final gather"; one process here, so direct
  # peer copies): every shard to the first handle's GPU, one hop each over xGMI, then ONE copy to the host
  Xall, Uall = zeros(n, N, B), zeros(m, N, B)
  if length(shards) > 1
    hs = Ptr{Cvoid}[h for (_, _, h) in shards]
    gusto_check(ccall((:gusto_gather_peer, libgusto_hip), Cint,
                      (Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cdouble}}, Ptr{Ptr{Cdouble}}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}),
                      hs[1], length(hs), hs, C_NULL, C_NULL, Xall, Uall, C_NULL), hs[1], "gather_peer")
  end
  for (b0, b1, h) in shards
    Bs = b1 - b0 + 1
    gusto_check(ccall((:gusto_wait, libgusto_hip), Cint, (Ptr{Cvoid},), h), h, "wait")
    X, U = zeros(n, N, Bs), zeros(m, N, Bs)
    if length(shards) > 1
      X .= Xall[:, :, b0:b1]; U .= Uall[:, :, b0:b1]
    else
      gusto_check(ccall((:gusto_get_traj, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), h, X, U), h, "get_traj")
    end
    its, conv, succ, stop = zeros(Cint, Bs), zeros(Cint, Bs), zeros(Cint, Bs), zeros(Cint, Bs)
    gusto_check(ccall((:gusto_get_status, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cvoid}),
                      h, its, conv, succ, stop, C_NULL), h, "get_status")
    duals = zeros(n, Bs)
    gusto_check(ccall((:gusto_get_dual, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), h, duals), h, "get_dual")
    Hs = gusto_histories(h, Bs)                      # one device -> host copy per shard
    msec = Ref{Cdouble}(0.)
    ccall((:gusto_last_solve_ms, libgusto_hip), Cint, (Ptr{Cvoid}, Ref{Cdouble}), h, msec)
    for b in b0:b1
      j = b - b0 + 1
      SCPP = SCPProblem(TOPs[b])
      SCPP.param.alg = SCPParam_GuSTO(model)
      SCPS = SCPSolution(SCPP, Trajectory(X[:, :, j], U[:, :, j], TOPs[b].tf_guess))
      gusto_fill_solution!(SCPS, SCPP.param.alg, Hs, j, its[j], conv[j], succ[j], stop[j], duals[:, j])
      SCPS.total_time = msec[] / 1e3 / Bs
      SCPS.iter_elapsed_times = vcat(0., fill(SCPS.total_time / max(1, its[j]), its[j]))
      SCPP.param.obstacle_toggle_distance = SCPP.param.alg.Δ_vec[end] / 8 + model.clearance
      TOSs[b].traj, TOSs[b].SCPS = SCPS.traj, SCPS
    end
    ccall((:gusto_destroy, libgusto_hip), Cint, (Ptr{Cvoid},), h)
  end
  nothing
end

# ---- TrajOpt behind the same seam: solve_SCP!(TOS, TOP, solve_trajopt_hip!, init_traj_straightline, "hip") -----------------
# positional signature of solve_trajopt_jump! (src/scp/scp_trajopt.jl:33); FreeflyerSE2 and AstrobeeSE3
struct GustoTrajOptParams      # gusto_trajopt_params
  mu0::Cdouble; s0::Cdouble; c::Cdouble; tau_plus::Cdouble; tau_minus::Cdouble; k::Cdouble; ftol::Cdouble; xtol::Cdouble; ctol::Cdouble
  max_penalty_iteration::Cint; max_convex_iteration::Cint; max_trust_iteration::Cint
end

mutable struct GustoTrajOptHistory  # gusto_trajopt_history
  hist_cap::Cint
  n_solves::Ptr{Cint}; n_mu::Ptr{Cint}; n_xtol::Ptr{Cint}; n_ftol::Ptr{Cint}; n_ctol::Ptr{Cint}
  rho_vec::Ptr{Cdouble}; s_vec::Ptr{Cdouble}; mu_vec::Ptr{Cdouble}; xtol_vec::Ptr{Cdouble}; ftol_vec::Ptr{Cdouble}; ctol_vec::Ptr{Cdouble}
  J_true::Ptr{Cdouble}; J_full::Ptr{Cdouble}; convergence_measure::Ptr{Cdouble}
  solver_status::Ptr{Cint}; ipm_iters::Ptr{Cint}
end

function solve_trajopt_hip!(SCPS::SCPSolution, SCPP::SCPProblem, solver="hip", max_iter=125, force=false; device=0, kwarg...)
  model, N = SCPP.PD.model, SCPP.N
  n, m = model.x_dim, model.u_dim
  SCPP.param.alg = SCPParam_TrajOpt(model)                       # :44
  a = SCPP.param.alg
  tp = GustoTrajOptParams(a.mu0, a.s0, a.c, a.τ_plus, a.τ_minus, a.k, a.ftol, a.xtol, a.ctol,
                          a.max_penalty_iteration, a.max_convex_iteration, a.max_trust_iteration)
  cap = 2 * a.max_penalty_iteration * a.max_convex_iteration * a.max_trust_iteration + 16
  href = Ref{Ptr{Cvoid}}(C_NULL)
  gusto_check(ccall((:gusto_create_trajopt, libgusto_hip), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Cint),
                    href, gusto_model_id(model), N, 1, cap, device), href[], "create_trajopt")
  h = href[]
  gusto_check(ccall((:gusto_set_params, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{GustoModelParams}),
                    h, C_NULL, gusto_model_params(SCPP.PD.robot, model)), h, "set_params")
  gusto_check(ccall((:gusto_set_trajopt_params, libgusto_hip), Cint, (Ptr{Cvoid}, Ref{GustoTrajOptParams}), h, tp), h, "set_trajopt_params")
  boxes, spheres = gusto_env_tables(SCPP.PD.env)
  gusto_check(ccall((:gusto_set_env, libgusto_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}),
                    h, length(boxes) ÷ 6, boxes, length(spheres) ÷ 4, spheres), h, "set_env")
  lo, hi = gusto_goal_bounds(SCPP.PD.goal_set, n, SCPP.tf_guess)
  gusto_check(ccall((:gusto_set_problems, libgusto_hip), Cint,
                    (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                    h, 1, Float64.(SCPP.PD.x_init), lo, hi, [Float64(SCPP.tf_guess)], Matrix{Float64}(SCPS.traj.X), Matrix{Float64}(SCPS.traj.U)), h, "set_problems")
  t0 = time_ns()
  gusto_check(ccall((:gusto_solve_trajopt, libgusto_hip), Cint, (Ptr{Cvoid}, Cint), h, max_iter), h, "solve_trajopt")
  elapsed = (time_ns() - t0) / 10^9
  X, U = zeros(n, N), zeros(m, N)
  gusto_check(ccall((:gusto_get_traj, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), h, X, U), h, "get_traj")
  SCPS.traj.X, SCPS.traj.U = X, U
  its, conv, stop = zeros(Cint, 1), zeros(Cint, 1), zeros(Cint, 1)
  gusto_check(ccall((:gusto_get_status, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}, Ptr{Cvoid}, Ptr{Cint}, Ptr{Cvoid}),
                    h, its, conv, C_NULL, stop, C_NULL), h, "get_status")
  d() = zeros(Cdouble, cap); ns, nm, nx, nf, nc = (zeros(Cint, 1) for _ in 1:5)
  rho, sv, muv, xt, ft, ct, Jt, Jf, cm = (d() for _ in 1:9)
  sol, ipi = zeros(Cint, cap), zeros(Cint, cap)
  hist = GustoTrajOptHistory(cap, pointer(ns), pointer(nm), pointer(nx), pointer(nf), pointer(nc), pointer(rho), pointer(sv), pointer(muv),
                             pointer(xt), pointer(ft), pointer(ct), pointer(Jt), pointer(Jf), pointer(cm), pointer(sol), pointer(ipi))
  GC.@preserve ns nm nx nf nc rho sv muv xt ft ct Jt Jf cm sol ipi begin
    gusto_check(ccall((:gusto_get_trajopt_history, libgusto_hip), Cint, (Ptr{Cvoid}, Ref{GustoTrajOptHistory}), h, hist), h, "get_trajopt_history")
  end
  S = Int(ns[1])
  SCPS.J_true, SCPS.J_full = Jt[1:S+1], Jf[1:S]
  SCPS.convergence_measure = vcat(0., cm[2:S+1])
  SCPS.solver_status = [GUSTO_SOLVER_STATUS[s+1] for s in sol[1:S+1]]
  SCPS.iterations, SCPS.converged = S, conv[1] != 0
  SCPS.total_time += elapsed
  SCPS.iter_elapsed_times = vcat(0., fill(SCPS.total_time / max(1, S), S))
  dual = zeros(n)
  gusto_check(ccall((:gusto_get_dual, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), h, dual), h, "get_dual")
  SCPS.dual = dual
  a.ρ_vec, a.s_vec, a.mu_vec = rho[1:S+1], sv[1:S+1], muv[1:nm[1]]
  a.xtol_vec, a.ftol_vec, a.ctol_vec = xt[1:nx[1]], ft[1:nf[1]], ct[1:nc[1]]
  SCPP.param.obstacle_toggle_distance = model.clearance + 1.     # :64
  ccall((:gusto_destroy, libgusto_hip), Cint, (Ptr{Cvoid},), h)
  nothing
end

# ---- solve_SCPshooting! (src/traj_opt.jl:4-45) for a batch on ONE handle ---------------------------------------------------
# Per round one gusto_shoot and one gusto_solve(h, 1, 0) over the problems still in their loop; gusto_set_active carries the
# per-problem loop condition `!SCPS.converged && SCPS.iterations < max_iter` (traj_opt.jl:23).  Mirrors
# gusto.jl_amd/host.py: solve_SCPshooting_batch (which the GPU tests hold bit for bit against the single-problem driver).
function solve_SCPshooting_batch!(TOSs::Vector, TOPs::Vector, init_method=init_traj_straightline; max_iter=30, device=0)
  TOP0 = TOPs[1]; model, N = TOP0.PD.model, TOP0.N
  n, m, B = model.x_dim, model.u_dim, length(TOPs)
  all(T -> typeof(T.PD.model) == typeof(model) && T.N == N, TOPs) ||
    error("solve_SCPshooting_batch!: all problems must share the model type and N")
  # ONE handle = one set of SCP and model parameters: a batch of different robots or thresholds is refused, not solved with problem 1's
  mp0 = gusto_model_params(TOP0.PD.robot, model)
  all(T -> isequal(gusto_model_params(T.PD.robot, T.PD.model), mp0) && T.fixed_final_time == TOP0.fixed_final_time, TOPs) ||
    error("solve_SCPshooting_batch!: all problems must share the model / robot parameters and the SCP parameters")
  alg0 = SCPParam_GuSTO(model)
  thr = SCPParam(model, TOP0.fixed_final_time).convergence_threshold
  sp = GustoScpParams(alg0.Δ0, alg0.ω0, alg0.ω_max, alg0.ε, alg0.ρ0, alg0.ρ1, alg0.β_succ, alg0.β_fail, alg0.γ_fail, thr)
  href = Ref{Ptr{Cvoid}}(C_NULL)
  gusto_check(ccall((:gusto_create, libgusto_hip), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Cint),
                    href, gusto_model_id(model), N, B, gusto_hist_cap(max_iter), device), href[], "create")
  h = href[]
  gusto_check(ccall((:gusto_set_params, libgusto_hip), Cint, (Ptr{Cvoid}, Ref{GustoScpParams}, Ref{GustoModelParams}),
                    h, sp, gusto_model_params(TOP0.PD.robot, model)), h, "set_params")
  envs = [gusto_env_tables(T.PD.env) for T in TOPs]
  nb = Cint[length(e[1]) ÷ 6 for e in envs]; ns = Cint[length(e[2]) ÷ 4 for e in envs]
  gusto_check(ccall((:gusto_set_env_batch, libgusto_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cdouble}),
                    h, B, nb, vcat((vec(e[1]) for e in envs)...), ns, vcat((vec(e[2]) for e in envs)...)), h, "set_env_batch")
  x0 = hcat((Float64.(T.PD.x_init) for T in TOPs)...)
  bounds = [gusto_goal_bounds(T.PD.goal_set, n, T.tf_guess) for T in TOPs]
  lo, hi = hcat(first.(bounds)...), hcat(last.(bounds)...)
  inits = [init_method(T) for T in TOPs]
  X0, U0 = cat((t.X for t in inits)..., dims=3), cat((t.U for t in inits)..., dims=3)
  gusto_check(ccall((:gusto_set_problems, libgusto_hip), Cint,
                    (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                    h, B, x0, lo, hi, Float64[T.tf_guess for T in TOPs], X0, U0), h, "set_problems")
  SCPPs = [SCPProblem(T) for T in TOPs]
  for P in SCPPs; P.param.alg = SCPParam_GuSTO(model); end
  SCPSs = [SCPSolution(SCPPs[b], inits[b]) for b in 1:B]
  stops = zeros(Cint, B)
  for b in 1:B
    TOSs[b].SCPS = SCPSs[b]
    TOSs[b].SS = ShootingSolution(ShootingProblem(TOPs[b], SCPSs[b]), deepcopy(inits[b]))
  end
  function scp_round!(live)       # solve_method!(SCPS, SCPP, solver, 1) of every live problem: ONE launch
    gusto_check(ccall((:gusto_set_active, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cint}), h, Cint.(live)), h, "set_active")
    gusto_check(ccall((:gusto_solve, libgusto_hip), Cint, (Ptr{Cvoid}, Cint, Cint), h, 1, 0), h, "solve")
    X, U = zeros(n, N, B), zeros(m, N, B)
    gusto_check(ccall((:gusto_get_traj, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), h, X, U), h, "get_traj")
    its, conv, succ = zeros(Cint, B), zeros(Cint, B), zeros(Cint, B)
    gusto_check(ccall((:gusto_get_status, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cvoid}),
                      h, its, conv, succ, stops, C_NULL), h, "get_status")
    duals = zeros(n, B)
    gusto_check(ccall((:gusto_get_dual, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), h, duals), h, "get_dual")
    Hs = gusto_histories(h, B)
    for b in findall(live)
      SCPSs[b].traj = Trajectory(X[:, :, b], U[:, :, b], TOPs[b].tf_guess)
      gusto_fill_solution!(SCPSs[b], SCPPs[b].param.alg, Hs, b, its[b], conv[b], succ[b], stops[b], duals[:, b])
    end
  end
  live = trues(B)
  scp_round!(live)
  for b in 1:B; push!(TOSs[b].SS.J_true, SCPSs[b].J_true[1]); end
  by_shooting = falses(B)
  while true
    # (stop reasons 2 / 3 / 4 -- failed subproblem, omega > omega_max, history full -- end a problem's loop: solve_method! would
    # return early for ever without counting an iteration, scp_gusto.jl:107-111,163-166)
    live = [!SCPSs[b].converged && SCPSs[b].iterations < max_iter && !(stops[b] in (2, 3, 4)) && !by_shooting[b] for b in 1:B]
    any(live) || break
    gusto_check(ccall((:gusto_set_active, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cint}), h, Cint.(live)), h, "set_active")
    t0 = time_ns()
    gusto_check(ccall((:gusto_shoot, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ref{GustoShootOpts}),
                      h, C_NULL, GustoShootOpts(4, 100, 1e-3, 0)), h, "shoot")     # seeds = SCPS.dual of every problem
    st, it, res, p0 = zeros(Cint, B), zeros(Cint, B), zeros(B), zeros(n, B)
    X, U = zeros(n, N, B), zeros(m, N, B)
    gusto_check(ccall((:gusto_get_shoot, libgusto_hip), Cint,
                      (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), h, st, it, res, p0, X, U), h, "get_shoot")
    el = (time_ns() - t0) / 10^9 / count(live)
    for b in findall(live)
      SS, SCPS = TOSs[b].SS, SCPSs[b]
      SP = ShootingProblem(TOPs[b], SCPS)
      if st[b] == 1
        new_traj = Trajectory(X[:, :, b], U[:, :, b], SP.tf)
        push!(SS.prob_status, :Optimal); push!(SS.J_true, cost_true(new_traj, new_traj, SP))
        push!(SS.convergence_measure, convergence_metric(new_traj, SS.traj, SP))
        copy!(SS.traj, new_traj)
      else
        push!(SS.prob_status, :Diverged); push!(SS.J_true, NaN); push!(SS.convergence_measure, NaN)
      end
      push!(SS.iter_elapsed_times, el)
      cm = SS.convergence_measure[end-1:end]
      if SCPS.iterations > 2 && !any(isnan, cm) && sum(cm) <= thr      # traj_opt.jl:30
        SS.converged = true; by_shooting[b] = true
      end
    end
    live = live .& .!by_shooting
    any(live) && scp_round!(live)
  end
  for b in 1:B
    copy!(TOSs[b].traj, by_shooting[b] ? TOSs[b].SS.traj : SCPSs[b].traj)
    TOSs[b].total_time = SCPSs[b].total_time + sum(TOSs[b].SS.iter_elapsed_times)
  end
  ccall((:gusto_destroy, libgusto_hip), Cint, (Ptr{Cvoid},), h)
  nothing
end

# ---- TrajOpt for a batch, one handle per GPU, shards side by side (gusto_solve_trajopt_async + gusto_wait) -----------------
# The batch counterpart of solve_trajopt_hip! (mirrors host.py: solve_SCP_batch(..., solve_trajopt_hip, devices = [...])):
# contiguous shards of ceil(B/G) problems, every shard's ONE launch enqueued before the first is waited for.
function solve_trajopt_batch!(TOSs::Vector, TOPs::Vector, init_method=init_traj_straightline; max_iter=125, device=0, devices=nothing)
  TOP0 = TOPs[1]; model, N = TOP0.PD.model, TOP0.N
  n, m, B = model.x_dim, model.u_dim, length(TOPs)
  all(T -> typeof(T.PD.model) == typeof(model) && T.N == N, TOPs) ||
    error("solve_trajopt_batch!: all problems must share the model type and N")
  a = SCPParam_TrajOpt(model)
  tp = GustoTrajOptParams(a.mu0, a.s0, a.c, a.τ_plus, a.τ_minus, a.k, a.ftol, a.xtol, a.ctol,
                          a.max_penalty_iteration, a.max_convex_iteration, a.max_trust_iteration)
  cap = 2 * a.max_penalty_iteration * a.max_convex_iteration * a.max_trust_iteration + 16
  devs = devices === nothing ? [device] : collect(devices)
  G = length(devs); per = cld(B, G)
  x0 = hcat((Float64.(T.PD.x_init) for T in TOPs)...)
  bounds = [gusto_goal_bounds(T.PD.goal_set, n, T.tf_guess) for T in TOPs]
  lo, hi = hcat(first.(bounds)...), hcat(last.(bounds)...)
  inits = [init_method(T) for T in TOPs]
  X0, U0 = cat((t.X for t in inits)..., dims=3), cat((t.U for t in inits)..., dims=3)
  envs = [gusto_env_tables(T.PD.env) for T in TOPs]
  shards = Tuple{Int,Int,Ptr{Cvoid}}[]
  for (r, dv) in enumerate(devs)
    b0, b1 = min(B, (r - 1) * per) + 1, min(B, r * per)
    b1 < b0 && continue
    href = Ref{Ptr{Cvoid}}(C_NULL)
    gusto_check(ccall((:gusto_create_trajopt, libgusto_hip), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Cint),
                      href, gusto_model_id(model), N, b1 - b0 + 1, cap, dv), href[], "create_trajopt")
    h = href[]
    gusto_check(ccall((:gusto_set_params, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{GustoModelParams}),
                      h, C_NULL, gusto_model_params(TOP0.PD.robot, model)), h, "set_params")
    gusto_check(ccall((:gusto_set_trajopt_params, libgusto_hip), Cint, (Ptr{Cvoid}, Ref{GustoTrajOptParams}), h, tp), h, "set_trajopt_params")
    nb = Cint[length(e[1]) ÷ 6 for e in envs[b0:b1]]; ns = Cint[length(e[2]) ÷ 4 for e in envs[b0:b1]]
    gusto_check(ccall((:gusto_set_env_batch, libgusto_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cdouble}),
                      h, b1 - b0 + 1, nb, vcat((vec(e[1]) for e in envs[b0:b1])...), ns, vcat((vec(e[2]) for e in envs[b0:b1])...)), h, "set_env_batch")
    gusto_check(ccall((:gusto_set_problems, libgusto_hip), Cint,
                      (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                      h, b1 - b0 + 1, x0[:, b0:b1], lo[:, b0:b1], hi[:, b0:b1], Float64[T.tf_guess for T in TOPs[b0:b1]],
                      X0[:, :, b0:b1], U0[:, :, b0:b1]), h, "set_problems")
    gusto_check(ccall((:gusto_solve_trajopt_async, libgusto_hip), Cint, (Ptr{Cvoid}, Cint), h, max_iter), h, "solve_trajopt_async")
    push!(shards, (b0, b1, h))
  end
  for (b0, b1, h) in shards
    Bs = b1 - b0 + 1
    gusto_check(ccall((:gusto_wait, libgusto_hip), Cint, (Ptr{Cvoid},), h), h, "wait")
    X, U = zeros(n, N, Bs), zeros(m, N, Bs)
    gusto_check(ccall((:gusto_get_traj, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), h, X, U), h, "get_traj")
    its, conv = zeros(Cint, Bs), zeros(Cint, Bs)
    gusto_check(ccall((:gusto_get_status, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                      h, its, conv, C_NULL, C_NULL, C_NULL), h, "get_status")
    duals = zeros(n, Bs)
    gusto_check(ccall((:gusto_get_dual, libgusto_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), h, duals), h, "get_dual")
    Jt, Jf, cm = zeros(cap, Bs), zeros(cap, Bs), zeros(cap, Bs)      # rows of `cap` entries, problem slowest
    nsol = zeros(Cint, Bs)
    hist = GustoTrajOptHistory(cap, pointer(nsol), C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL,
                               pointer(Jt), pointer(Jf), pointer(cm), C_NULL, C_NULL)
    GC.@preserve nsol Jt Jf cm begin
      gusto_check(ccall((:gusto_get_trajopt_history, libgusto_hip), Cint, (Ptr{Cvoid}, Ref{GustoTrajOptHistory}), h, hist), h, "get_trajopt_history")
    end
    msec = Ref{Cdouble}(0.)
    ccall((:gusto_last_solve_ms, libgusto_hip), Cint, (Ptr{Cvoid}, Ref{Cdouble}), h, msec)
    for b in b0:b1
      j = b - b0 + 1; S = Int(nsol[j])
      SCPP = SCPProblem(TOPs[b]); SCPP.param.alg = SCPParam_TrajOpt(model)
      SCPS = SCPSolution(SCPP, Trajectory(X[:, :, j], U[:, :, j], TOPs[b].tf_guess))
      SCPS.J_true, SCPS.J_full = Jt[1:S+1, j], Jf[1:S, j]
      SCPS.convergence_measure = vcat(0., cm[2:S+1, j])
      SCPS.iterations, SCPS.converged, SCPS.dual = S, conv[j] != 0, duals[:, j]
      SCPS.total_time = msec[] / 1e3 / Bs
      TOSs[b].traj, TOSs[b].SCPS = SCPS.traj, SCPS
    end
    ccall((:gusto_destroy, libgusto_hip), Cint, (Ptr{Cvoid},), h)
  end
  nothing
end
