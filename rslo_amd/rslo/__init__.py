"""Host-side mirror of the reference's `rslo` package, restricted to the hot path
(SURVEY.md section 8): same module paths, class names, constructor kwargs and state-dict keys."""
