R=$GRAFT_REPO_ROOT; cd $R
python -c "
import cProfile, pstats, sys, runpy
sys.argv=['tools/profile_train_groups.py','12']
cProfile.run('runpy.run_path(\"tools/profile_train_groups.py\", run_name=\"__main__\")', '/tmp/prof.out')
p=pstats.Stats('/tmp/prof.out'); p.sort_stats('tottime').print_stats(38)
" 2>&1 | tail -60
