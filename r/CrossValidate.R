#' Drop-in replacement for parEBEN::CrossValidate (R/CrossValidate.R:61-117) whose global search
#' evaluates the nFolds x alpha x lambda grid on MI355X GPUs through libpareben_hip.so.
#' Same leading arguments, same returned list; AssignToFolds() and LocalSearch() are the package's own.
#' No foreach backend is needed for search = "global".
#'
#' Trailing, optional (SURVEY.md 8(b)): nAlpha / nLambda -- grid sizes (the reference hard-wires 20 x 20,
#' R/BuildGrid.R:38,44; other sizes keep its lambda range and alpha spacing); nGPU -- 1 = one GPU (`device`),
#' 0 = every visible GPU, k = the first k: one host thread per GPU inside the library and a single RCCL
#' all-gather of the per-cell errors (include/pareben_hip.h: pareben_cv_grid_multi); device -- GPU for nGPU = 1.
#' @useDynLib parEBEN, .registration = TRUE
CrossValidate <- function(BASIS, Target, nFolds, foldId = 0, Epis = "no", prior = "gaussian",
                          search = "global", nAlpha = 20L, nLambda = 20L, nGPU = 1L, device = 0L){
  if(search != "global") return(LocalSearch(BASIS, Target, nFolds, Epis, foldId, prior))
  storage.mode(BASIS) <- "double"
  ParameterGrid <- BuildGridGPU(BASIS, Target, nFolds, Epis, nAlpha, nLambda, device)
  folds <- AssignToFolds(BASIS, nFolds)           # what TestModel() uses for every fit (R/TestModel.R:9)
  res <- .Call(pareben_cv_grid_R, BASIS, as.double(Target), as.integer(folds), as.integer(nFolds),
               as.double(ParameterGrid$alpha), as.double(ParameterGrid$lambda),
               as.integer(Epis == "yes"), as.integer(prior != "gaussian"), as.integer(nGPU), as.integer(device))
  stopped <- sum(bitwAnd(res$status, 8L) != 0L)
  if(stopped > 0) warning(stopped, " of ", length(res$status), " fits were stopped early (status bit 8) and score NA")
  nCells <- nrow(ParameterGrid)
  detail <- data.frame(foldId = rep(1:nFolds, nCells),
                       alpha  = rep(ParameterGrid$alpha,  each = nFolds),
                       lambda = rep(ParameterGrid$lambda, each = nFolds))
  if(prior == "gaussian"){
    detail$MSE <- as.vector(res$fold_err)
    Error <- detail %>% group_by(alpha, lambda) %>%
      summarise(SE = sd(MSE)/sqrt(max(foldId)), MSE = mean(MSE))
    index <- which.min(Error$MSE)
  }else{
    detail$logL <- as.vector(res$fold_err)
    Error <- detail %>% group_by(alpha, lambda) %>%
      summarise(SE = sd(logL)/sqrt(max(foldId)), Likelihood = -mean(logL))
    index <- which.min(Error$Likelihood)          # the reference indexes Error$MSE here (SURVEY.md Q8)
  }
  list(Results.Detail = detail, Results.Summary = Error,
       lambda.optimal = Error[index,]$lambda, alpha.optimal = Error[index,]$alpha)
}

#' BuildGrid() (R/BuildGrid.R:34-52) with the pairwise pass of GetLambdaMax (:21-30, an interpreted O(n K^2) double
#' loop) on the GPU, and optional grid sizes.  nAlpha = nLambda = 20 reproduces the reference grid exactly.
BuildGridGPU <- function(BASIS, Target, nFolds, Epis = "no", nAlpha = 20L, nLambda = 20L, device = 0L){
  lambda_Max <- GetLambdaMax(BASIS, Target, "no")                       # main-effect pass + log(1.1) floor, :9-19
  if(Epis == "yes"){
    pair <- .Call(pareben_lambda_max_pairs_R, BASIS, as.double(Target), as.integer(device))
    if(pair > lambda_Max) lambda_Max <- pair
  }
  lambda_Max <- lambda_Max * 10
  lambda_Min <- log(0.001 * lambda_Max)
  step <- (log(lambda_Max) - lambda_Min)/(nLambda - 1)
  Lambda <- exp(seq(from = log(lambda_Max), to = lambda_Min, by = -step))
  Alpha <- if(nAlpha == 20L) seq(from = 1, to = 0.05, by = -0.05) else seq(from = 1, to = 1/nAlpha, by = -1/nAlpha)
  as.data.frame(expand.grid(alpha = Alpha, lambda = Lambda))
}

#' The refit every user script runs after the search (tests/CrossValidate-test.R:23): EBEN::EBelasticNet.Gaussian with
#' its .C(...) line (EBEN_orig/R/EBelasticNet.Gaussian.R:16-51) swapped for the matching .Call; the row filter, the
#' main / pair ordering and the t / p columns are EBEN's own R code and stay as they are.
EBelasticNet.Gaussian.GPU <- function(BASIS, Target, lambda, alpha, Epis = "no", verbose = 0, device = 0L){
  storage.mode(BASIS) <- "double"
  out <- .Call(pareben_fit_gaussian_R, BASIS, as.double(Target), as.double(lambda), as.double(alpha),
               as.integer(Epis == "yes"), as.integer(device))
  out        # Beta (K x 4 | K(K+1)/2 x 5), WaldScore, Intercept, residual: what `output` holds after .C(...) in :16-51
}
