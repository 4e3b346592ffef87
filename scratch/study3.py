from nb import *
env=P.freeflyer_env()
o=go.Oracle(go.FREEFLYER_SE2,200,boxes=env)
o.set_problem(P.FREEFLYER_X_INIT,P.FREEFLYER_X_GOAL,P.FREEFLYER_X_GOAL,P.FREEFLYER_TF)
X,U=o.init_straightline()
for c in range(2):
  for i in range(14):
    d=[o.signed_distance(c,X[k,:2],i)[0] for k in range(200)]
    print(c,i,'min d %.4f at k=%d; n(d<0)=%d n(d<0.425)=%d'%(min(d),int(np.argmin(d)),sum(x<0 for x in d),sum(x<0.425 for x in d)))
