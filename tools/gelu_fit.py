import numpy as np
from scipy.special import erf
from scipy.optimize import minimize
x=np.linspace(-9,9,200001)
g=0.5*x*(1+erf(x/np.sqrt(2)))
def approx(c,x):
    x2=x*x
    p=x*(c[0]+x2*(c[1]+x2*(c[2] if len(c)>2 else 0)+ (x2*x2*c[3] if len(c)>3 else 0)))
    return x/(1+np.exp(-p))
def loss(c): return np.max(np.abs(approx(c,x)-g))
for n in (2,3,4):
    c0=[1.5957691216,0.0713548163]+[0.0]*(n-2)
    best=None
    for it in range(6):
        r=minimize(loss,c0,method='Nelder-Mead',options={'xatol':1e-12,'fatol':1e-14,'maxiter':20000})
        c0=r.x
    print(n,r.x,loss(r.x))
