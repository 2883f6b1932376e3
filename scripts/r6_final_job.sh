cd $GRAFT_REPO_ROOT
bash scripts/gpu_job.sh r6_29 bench:c3 steady:c3 prof:c3 pmc:c3 pmc:c2 pmc:c4 pmc:c2is512 pmc:c4kl pmc:c5 pmc:c4sc pmc:c2is256 pmc:c4is steady:c2 steady:c4 steady:c4kl steady:c5 steady:c4sc steady:c2is steady:c2is256 steady:c4is steady:c2is512 prof:c2 prof:c4
