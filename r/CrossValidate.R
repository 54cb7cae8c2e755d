#' Drop-in replacement for parEBEN::CrossValidate (R/CrossValidate.R:61-117) whose global search
#' evaluates the nFolds x alpha x lambda grid on MI355X GPUs through libpareben_hip.so.
#' Same arguments, same returned list; BuildGrid(), AssignToFolds() and LocalSearch() are the
#' package's own.  No foreach backend is needed for search = "global".
CrossValidate <- function(BASIS, Target, nFolds, foldId = 0, Epis = "no", prior = "gaussian",
                          search = "global", device = 0L){
  if(search != "global") return(LocalSearch(BASIS, Target, nFolds, Epis, foldId, prior))
  ParameterGrid <- BuildGrid(BASIS, Target, nFolds, Epis)
  folds <- AssignToFolds(BASIS, nFolds)           # what TestModel() uses for every fit (R/TestModel.R:9)
  storage.mode(BASIS) <- "double"
  res <- .Call("pareben_cv_grid_R", BASIS, as.double(Target), as.integer(folds), as.integer(nFolds),
               as.double(ParameterGrid$alpha), as.double(ParameterGrid$lambda),
               as.integer(Epis == "yes"), as.integer(prior != "gaussian"), as.integer(device))
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
